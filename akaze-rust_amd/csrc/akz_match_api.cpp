// C ABI of libakaze_hip.so, part 4: descriptor_match (pair, multi-set, both directions), match_features.
#include "akz_ctx.hpp"

extern "C" {
// ---------------------------------------------------------------------------------------------
// descriptor_match
// ---------------------------------------------------------------------------------------------
// rows_le_61: every row uses at most its first 61 bytes (M-LDB descriptors: 486 bits; the matrix-core kernel keeps the
// train rows' bit counts in the K-columns of bytes 61..63 and ignores whatever those bytes hold)
static int match_device_impl(akz_ctx* c, const uint8_t* d_d0, uint64_t n0, const uint8_t* d_d1, uint64_t n1,
                             uint64_t distance_threshold, double lowes_ratio, akz_match* d_out, uint64_t* d_n_out,
                             bool rows_le_61) {
    AKZ_TRY(bind(c, true, false));  // (the matcher shares nothing with a finish half that may be running on the context's thread)
    if (!d_out || !d_n_out || (n0 && !d_d0) || (n1 && !d_d1) || n0 > 0x7fffffffull || n1 > 0x7fffffffull) {
        set_error("descriptor_match: bad arguments");
        return AKZ_ERR_INVALID_ARG;
    }
    const uint32_t thr = (uint32_t)std::min<uint64_t>(distance_threshold, 0x7fffffffull);
    // the scan runs on the matrix cores (integer GEMM on the unpacked bits, identical records): 24 us against 29 for the
    // popcount kernel at 128 x 128, 35 against 190 at 1024 x 1024, 3.0 ms against 9.9 at 90 K x 90 K; mode 0 keeps the
    // popcount kernel selectable
    const bool mfma = n0 && n1 && c->match_mode != 0 && rows_le_61;
    const uint32_t chunks = mfma ? launch::match_mfma_chunks((uint32_t)n0, (uint32_t)n1, c->dbg_pair_chunks)
                                 : launch::match_num_chunks((uint32_t)n0, (uint32_t)n1);
    AKZ_TRY(ensure(c, c->match_rec, std::max<uint64_t>(1, n0) * (chunks + 1) * sizeof(MatchRec)));
    MatchRec* rec = (MatchRec*)c->match_rec.p;  // [chunk][query], then the merged records
    if (mfma) {
        const uint32_t q_rows = launch::match_mfma_rows((uint32_t)n0, true), t_rows = launch::match_mfma_rows((uint32_t)n1, false);
        AKZ_TRY(ensure(c, c->mm_q8, (size_t)q_rows * 512));
        AKZ_TRY(ensure(c, c->mm_t8, (size_t)t_rows * 512));
        AKZ_TRY(ensure(c, c->mm_pop, ((size_t)2 * q_rows + t_rows) * sizeof(uint32_t)));
        uint32_t* qpop = (uint32_t*)c->mm_pop.p;
        uint32_t* bound = qpop + q_rows;
        uint32_t* tpop = bound + q_rows;
        const bool fp4 = c->match_mode >= 2;
        launch::unpack_pair(c->stream, d_d0, (uint32_t)n0, q_rows, (uint8_t*)c->mm_q8.p, qpop, bound, thr, d_d1, (uint32_t)n1, t_rows,
                            (uint8_t*)c->mm_t8.p, tpop, fp4);
        launch::match_mfma(c->stream, (const uint8_t*)c->mm_q8.p, qpop, (uint32_t)n0, (const uint8_t*)c->mm_t8.p,
                           (uint32_t)n1, thr, bound, chunks, rec, fp4);
    } else {
        launch::match(c->stream, d_d0, (uint32_t)n0, d_d1, (uint32_t)n1, thr, rows_le_61, chunks, rec);
    }
    // more than a few chunks: a parallel merge first (the compaction is ONE workgroup, i.e. one compute unit's load path:
    // folding 11 chunks of 11 K queries there took 35 us against 5 for the 44-workgroup merge)
    // Query sets of a few workgroups and more: merge, ratio test and ordered compaction in ONE launch (k_match_merge_compact:
    // 44 workgroups for a 4K frame's 11 K rows); tiny ones keep the single-workgroup compaction
    // k_match_merge_compact is a chained look-back: a workgroup publishes its count, then waits for the counts of the
    // workgroups before it.  Two conditions keep that free of deadlock and of cross-talk, and both are enforced HERE:
    //   * every workgroup of the grid is resident at once (progress never depends on the order of dispatch): at most 1 024
    //     workgroups of 256 threads -- four per compute unit -- i.e. query sets of up to 262 144 rows; larger sets take the
    //     merge + single-workgroup compaction below;
    //   * match_state (told apart by epoch) belongs to ONE stream: this function, like every user of the context's matcher
    //     scratch (match_rec, mm_q8, mm_t8, mm_pop), enqueues on c->stream only -- launches of two epochs never overlap.
    if (n0 >= gates::kMergeCompactMinRows && n0 <= gates::kMergeCompactMaxRows) {
        const size_t need = launch::match_merge_compact_state_bytes((uint32_t)n0);
        if (c->match_state.bytes < need) {
            AKZ_TRY(ensure(c, c->match_state, need * 2));
            AKZ_HIP_TRY(hipMemsetAsync(c->match_state.p, 0, c->match_state.bytes, c->stream));
        }
        if (++c->match_epoch == 0) ++c->match_epoch;
        launch::match_merge_compact(c->stream, rec, (uint32_t)n0, chunks, thr, lowes_ratio * lowes_ratio, d_out,
                                    (unsigned long long*)d_n_out, c->match_state.p, c->match_epoch);
        AKZ_HIP_TRY(hipGetLastError());
        return AKZ_OK;
    }
    const bool premerge = chunks > gates::kPremergeChunks;
    if (premerge) launch::match_merge(c->stream, rec, (uint32_t)n0, chunks, thr, rec + (size_t)chunks * n0);
    launch::match_compact(c->stream, premerge ? rec + (size_t)chunks * n0 : rec, (uint32_t)n0, premerge ? 1u : chunks, thr,
                          lowes_ratio * lowes_ratio, d_out, (unsigned long long*)d_n_out);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}

int akz_descriptor_match_device(akz_ctx* c, const uint8_t* d_d0, uint64_t n0, const uint8_t* d_d1, uint64_t n1,
                                uint64_t distance_threshold, double lowes_ratio, akz_match* d_out,
                                uint64_t* d_n_out) {
    return match_device_impl(c, d_d0, n0, d_d1, n1, distance_threshold, lowes_ratio, d_out, d_n_out, true);
}

// One query set against several train sets in ONE matrix-core launch (all-pairs matching: a query image against
// the descriptor sets of all other images).  A pair of 11 K-row sets alone runs at 1.2 T pairs/s, a launch over
// many sets at the rate of one large product (3 T pairs/s), and every set is unpacked once per call.
// set_first (optional): first row of set k inside d_train (sets anywhere in one block of rows, e.g. a gather's); without
// it the sets follow each other
static int match_sets_impl(akz_ctx* c, const uint8_t* d_q, uint64_t n0, const uint8_t* d_train, const uint64_t* set_rows,
                           uint64_t n_sets, uint64_t distance_threshold, double lowes_ratio, akz_match* d_out, uint64_t* d_n_out,
                           akz_match* d_out_cols, uint64_t* d_n_cols, const uint64_t* set_first = nullptr, int side = 0) {
    AKZ_TRY(bind(c, true, false));
    const bool cols = d_n_cols != nullptr;  // the opposite direction too: every set's rows against the query set
    if ((n0 && !d_out) || !d_n_out || (n_sets && !set_rows) || (n0 && !d_q) || n0 > 0x7fffffffull || n_sets > 65535) {
        set_error("descriptor_match_sets: bad arguments");
        return AKZ_ERR_INVALID_ARG;
    }
    uint64_t total_rows = 0, last_row = 0;
    std::vector<uint64_t> first(n_sets);  // first row of every set in d_train
    for (uint64_t k = 0; k < n_sets; ++k) {
        first[(size_t)k] = set_first ? set_first[k] : total_rows;
        total_rows += set_rows[k];
        last_row = std::max(last_row, first[(size_t)k] + set_rows[k]);
    }
    if ((total_rows && !d_train) || total_rows > 0x7fffffffull || last_row > 0xffffffffull) {
        set_error("descriptor_match_sets: bad train sets");
        return AKZ_ERR_INVALID_ARG;
    }
    if (n_sets == 0) return AKZ_OK;
    if (cols && total_rows && !d_out_cols) {
        set_error("descriptor_match_sets_mutual: null output for the opposite direction");
        return AKZ_ERR_INVALID_ARG;
    }
    if (n0 == 0) {  // (no query rows: nothing can match in either direction)
        AKZ_HIP_TRY(hipMemsetAsync(d_n_out, 0, n_sets * sizeof(uint64_t), c->stream));
        if (cols) AKZ_HIP_TRY(hipMemsetAsync(d_n_cols, 0, n_sets * sizeof(uint64_t), c->stream));
        return AKZ_OK;
    }
    if (cols && c->match_mode < 2) {  // both directions ride on the FP4 kernel only: the other kernels match direction by direction
        uint64_t off = 0;
        for (uint64_t k = 0; k < n_sets; ++k) {
            if (set_rows[k] == 0) AKZ_HIP_TRY(hipMemsetAsync(d_n_cols + k, 0, sizeof(uint64_t), c->stream));
            else
                AKZ_TRY(match_device_impl(c, d_train + first[(size_t)k] * 64, set_rows[k], d_q, n0, distance_threshold, lowes_ratio,
                                          d_out_cols + off, d_n_cols + k, true));
            off += set_rows[k];
        }
    }
    if (c->match_mode == 0) {  // popcount kernel: set by set
        for (uint64_t k = 0; k < n_sets; ++k)
            AKZ_TRY(match_device_impl(c, d_q, n0, d_train + first[(size_t)k] * 64, set_rows[k], distance_threshold, lowes_ratio,
                                      d_out + k * n0, d_n_out + k, true));
        return AKZ_OK;
    }
    // (side 1: the finish stream and the second scratch set -- akz_match_all_pairs alternates; FP4 forms only)
    const bool alt = side == 1 && c->match_mode >= 2;
    if (alt) AKZ_TRY(ensure_aux(c));
    hipStream_t st = alt ? c->aux : c->stream;
    DevBuf &b_q8 = alt ? c->ms1.q8 : c->mm_q8, &b_t8 = alt ? c->ms1.t8 : c->mm_t8, &b_pop = alt ? c->ms1.pop : c->mm_pop;
    DevBuf &b_tab = alt ? c->ms1.tab : c->mm_tab, &b_cols = alt ? c->ms1.cols : c->mm_cols, &b_rec = alt ? c->ms1.rec : c->match_rec;
    void*& ring = alt ? c->ms1.ring : c->tab_ring;
    size_t& ring_bytes = alt ? c->ms1.ring_bytes : c->tab_ring_bytes;
    uint64_t& ring_next = alt ? c->ms1.ring_next : c->tab_ring_next;
    hipEvent_t* ring_ev = alt ? c->ms1.ring_ev : c->tab_ring_ev;
    const bool mutual = cols && c->match_mode >= 2;
    const uint32_t thr = (uint32_t)std::min<uint64_t>(distance_threshold, 0x7fffffffull);
    const uint32_t tr = launch::match_mfma_tile_rows();
    const uint32_t q_rows = launch::match_mfma_rows((uint32_t)n0, true);
    // padded train image: every set starts on a tile boundary
    std::vector<uint32_t> tiles;  // {first source row, valid rows} per tile
    // every set is cut into `cps` chunks (ascending rows; a short set leaves its last chunks empty) so that the
    // workgroups fill whole rounds of the chip; the chunks of a set share its pruning bounds
    // (chunks are the launch's gridDim.y: at most 65535 of them)
    const uint32_t cps = std::max(1u, std::min(launch::match_mfma_multi_chunks((uint32_t)n0, (uint32_t)n_sets,
                                                                               (uint32_t)((total_rows / n_sets + tr - 1) / tr), c->dbg_set_chunks),
                                               65535u / (uint32_t)n_sets));
    std::vector<launch::MatchChunkHost> chunks((size_t)n_sets * cps);
    std::vector<launch::MatchColSetHost> colsets(mutual ? (size_t)n_sets : 0);
    uint64_t src = 0;
    for (uint64_t k = 0; k < n_sets; ++k) {
        const uint32_t t0 = (uint32_t)(tiles.size() / 2), rows = (uint32_t)set_rows[k];
        if (mutual) colsets[(size_t)k] = launch::MatchColSetHost{t0 * tr, rows, (uint32_t)src};
        for (uint32_t r = 0; r < rows; r += tr) {
            tiles.push_back((uint32_t)(first[(size_t)k] + r));
            tiles.push_back(std::min(tr, rows - r));
        }
        const uint32_t t1 = (uint32_t)(tiles.size() / 2), per = (t1 - t0 + cps - 1) / cps;
        for (uint32_t j = 0; j < cps; ++j)
            chunks[k * cps + j] = launch::MatchChunkHost{std::min(t0 + j * per, t1), std::min(t0 + (j + 1) * per, t1), t0 * tr, rows,
                                                        (uint32_t)(k * q_rows), (uint32_t)(k * cps + j)};
        src += rows;
    }
    const uint32_t n_tiles = (uint32_t)(tiles.size() / 2), t_rows = std::max(1u, n_tiles) * tr;
    if ((uint64_t)n_sets * q_rows > 0x7fffffffull) {
        set_error("descriptor_match_sets: too many sets for this query set");
        return AKZ_ERR_INVALID_ARG;
    }
    // (both directions: the train image is also read as a QUERY image by the seed launch -- whole query blocks of rows)
    const uint32_t t_rows_q = mutual ? launch::match_mfma_rows(t_rows, true) : t_rows;
    AKZ_TRY(ensure(c, b_q8, (size_t)q_rows * 512));
    AKZ_TRY(ensure(c, b_t8, (size_t)t_rows_q * 512));
    AKZ_TRY(ensure(c, b_pop, ((size_t)q_rows * (1 + n_sets) + t_rows) * sizeof(uint32_t)));
    const size_t tab_tiles = std::max<size_t>(1, tiles.size()) * sizeof(uint32_t);
    const size_t tab_chunks = chunks.size() * sizeof(launch::MatchChunkHost), tab_cols = colsets.size() * sizeof(launch::MatchColSetHost);
    AKZ_TRY(ensure(c, b_tab, tab_tiles + tab_chunks + tab_cols));
    unsigned long long* cbest = nullptr;
    uint32_t *csecond = nullptr, *seed_bound = nullptr;
    MatchRec* seed_rec = nullptr;
    if (mutual) {  // per row of the padded train image: 8 + 4 (state) + 4 + 16 (seed launch) bytes
        AKZ_TRY(ensure(c, b_cols, (size_t)t_rows_q * 32));
        cbest = (unsigned long long*)b_cols.p;
        seed_rec = (MatchRec*)((char*)b_cols.p + (size_t)t_rows_q * 8);
        csecond = (uint32_t*)((char*)b_cols.p + (size_t)t_rows_q * 24);
        seed_bound = csecond + t_rows_q;
    }
    AKZ_TRY(ensure(c, b_rec, std::max<uint64_t>(1, n0) * chunks.size() * sizeof(MatchRec)));
    uint32_t* qpop = (uint32_t*)b_pop.p;
    uint32_t* bound = qpop + q_rows;
    uint32_t* tpop = bound + (size_t)n_sets * q_rows;
    uint32_t* d_tiles = (uint32_t*)b_tab.p;
    void* d_chunks = (char*)b_tab.p + tab_tiles;
    // The tables travel through a ring of pinned staging slots, so that the call returns without waiting for its copies
    // (a synchronisation here made every call of an all-pairs loop wait for the previous call's kernel).
    {
        constexpr int kRing = 4;
        const size_t need = tab_tiles + tab_chunks + tab_cols;
        if (ring_bytes < need) {
            AKZ_HIP_TRY(hipStreamSynchronize(st));
            if (ring) AKZ_HIP_TRY(hipHostFree(ring));
            ring = nullptr;
            ring_bytes = 0;
            AKZ_HIP_TRY(hipHostMalloc(&ring, (need + need / 2 + 4096) * kRing, hipHostMallocDefault));
            ring_bytes = need + need / 2 + 4096;
        }
        const int slot = (int)(ring_next++ % kRing);
        if (!ring_ev[slot]) AKZ_HIP_TRY(hipEventCreateWithFlags(&ring_ev[slot], hipEventDisableTiming));
        else AKZ_HIP_TRY(hipEventSynchronize(ring_ev[slot]));  // the copy that used this slot four calls ago
        char* stage = (char*)ring + (size_t)slot * ring_bytes;
        if (!tiles.empty()) std::memcpy(stage, tiles.data(), tiles.size() * sizeof(uint32_t));
        std::memcpy(stage + tab_tiles, chunks.data(), tab_chunks);
        if (tab_cols) std::memcpy(stage + tab_tiles + tab_chunks, colsets.data(), tab_cols);
        AKZ_HIP_TRY(hipMemcpyAsync(d_tiles, stage, need, hipMemcpyHostToDevice, st));
        AKZ_HIP_TRY(hipEventRecord(ring_ev[slot], st));
    }
    const bool fp4 = c->match_mode >= 2;
    launch::unpack_bits(st, d_q, (uint32_t)n0, q_rows, true, (uint8_t*)b_q8.p, qpop, bound, thr, (uint32_t)n_sets,
                        nullptr, fp4);
    if (n_tiles)
        launch::unpack_bits(st, d_train, 0, n_tiles * tr, false, (uint8_t*)b_t8.p, tpop, nullptr, 0, 0, d_tiles, fp4);
    if (mutual) {
        // the opposite direction rides along: seed the train rows' state from the first rows of the query set, then one pass
        launch::match_cols_seed(st, (const uint8_t*)b_q8.p, (uint32_t)n0, (const uint8_t*)b_t8.p, n_tiles * tr, thr,
                                seed_bound, seed_rec, cbest, csecond);
        launch::match_fp4_multi_mutual(st, (const uint8_t*)b_q8.p, (uint32_t)n0, (const uint8_t*)b_t8.p, d_chunks,
                                       (uint32_t)chunks.size(), thr, bound, (MatchRec*)b_rec.p, cbest, csecond);
        launch::match_compact_cols(st, cbest, csecond, (const char*)d_chunks + tab_chunks, (uint32_t)n_sets, thr,
                                   lowes_ratio * lowes_ratio, d_out_cols, (unsigned long long*)d_n_cols);
    } else {
        launch::match_mfma_multi(st, (const uint8_t*)b_q8.p, qpop, (uint32_t)n0, (const uint8_t*)b_t8.p, d_chunks,
                                 (uint32_t)chunks.size(), thr, bound, (MatchRec*)b_rec.p, fp4);
    }
    launch::match_compact_sets(st, (const MatchRec*)b_rec.p, (uint32_t)n0, (uint32_t)n_sets, cps, thr,
                               lowes_ratio * lowes_ratio, d_out, (unsigned long long*)d_n_out);
    AKZ_HIP_TRY(hipGetLastError());
    return AKZ_OK;
}
}  // extern "C"
// (akz_comm.cpp: the all-pairs match takes its sets where they lie in the gathered block)
int akz::match_sets_at(akz_ctx* c, const uint8_t* d_q, uint64_t n0, const uint8_t* d_rows, const uint64_t* set_first,
                       const uint64_t* set_rows, uint64_t n_sets, uint64_t distance_threshold, double lowes_ratio, akz_match* d_out,
                       uint64_t* d_n_out, akz_match* d_out_cols, uint64_t* d_n_cols, int side) {
    return match_sets_impl(c, d_q, n0, d_rows, set_rows, n_sets, distance_threshold, lowes_ratio, d_out, d_n_out, d_out_cols, d_n_cols,
                           set_first, side);
}
hipStream_t akz::match_side_stream(akz_ctx* c) {
    if (!c || c->match_mode < 2 || bind(c, true, false) != AKZ_OK || ensure_aux(c) != AKZ_OK) return nullptr;
    return c->aux;
}
extern "C" {
int akz_descriptor_match_sets_device(akz_ctx* c, const uint8_t* d_q, uint64_t n0, const uint8_t* d_train,
                                     const uint64_t* set_rows, uint64_t n_sets, uint64_t distance_threshold,
                                     double lowes_ratio, akz_match* d_out, uint64_t* d_n_out) {
    return match_sets_impl(c, d_q, n0, d_train, set_rows, n_sets, distance_threshold, lowes_ratio, d_out, d_n_out, nullptr, nullptr);
}
// Both directions of every (query set, train set k) block from ONE pass over the distances (hamming is symmetric): besides
// the lists of akz_descriptor_match_sets_device, the match list of set k's rows AS QUERIES against the query set as train
// (feature_matching.rs:23-94 with the two sets exchanged) goes to d_out_cols + (rows of the sets before k), its length to
// d_n_cols[k].  Identical to two akz_descriptor_match_device calls per block, at the matrix-core work of one.
int akz_descriptor_match_sets_mutual_device(akz_ctx* c, const uint8_t* d_q, uint64_t n0, const uint8_t* d_train,
                                            const uint64_t* set_rows, uint64_t n_sets, uint64_t distance_threshold,
                                            double lowes_ratio, akz_match* d_out, uint64_t* d_n_out, akz_match* d_out_cols,
                                            uint64_t* d_n_cols) {
    if (!d_n_cols) {
        set_error("descriptor_match_sets_mutual: null count output");
        return AKZ_ERR_INVALID_ARG;
    }
    return match_sets_impl(c, d_q, n0, d_train, set_rows, n_sets, distance_threshold, lowes_ratio, d_out, d_n_out, d_out_cols, d_n_cols);
}

int akz_descriptor_match(akz_ctx* c, const uint8_t* d0, uint64_t n0, const uint8_t* d1, uint64_t n1,
                         uint64_t desc_bytes, uint64_t distance_threshold, double lowes_ratio, akz_match* out,
                         uint64_t* n_out) {
    AKZ_TRY(bind(c, true, false));
    if (!n_out || desc_bytes == 0 || desc_bytes > 64 || (n0 && (!d0 || !out)) || (n1 && !d1)) {
        set_error("descriptor_match: bad arguments (desc_bytes must be 1..64)");
        return AKZ_ERR_INVALID_ARG;
    }
    *n_out = 0;
    if (n0 == 0) return AKZ_OK;
    auto pad = [&](const uint8_t* src, uint64_t cnt, std::vector<uint8_t>& dst) {
        dst.assign((size_t)std::max<uint64_t>(1, cnt) * 64, 0);
        for (uint64_t i = 0; i < cnt; ++i) std::memcpy(&dst[(size_t)i * 64], src + i * desc_bytes, desc_bytes);
    };
    std::vector<uint8_t> p0, p1;
    pad(d0, n0, p0);
    pad(d1, n1, p1);
    AKZ_TRY(ensure(c, c->match_a, p0.size()));
    AKZ_TRY(ensure(c, c->match_b, p1.size()));
    AKZ_TRY(ensure(c, c->match_out, n0 * sizeof(akz_match) + 64));
    AKZ_HIP_TRY(hipMemcpyAsync(c->match_a.p, p0.data(), p0.size(), hipMemcpyHostToDevice, c->stream));
    AKZ_HIP_TRY(hipMemcpyAsync(c->match_b.p, p1.data(), p1.size(), hipMemcpyHostToDevice, c->stream));
    akz_match* d_m = (akz_match*)((char*)c->match_out.p + 64);
    uint64_t* d_cnt = (uint64_t*)c->match_out.p;
    AKZ_TRY(match_device_impl(c, (const uint8_t*)c->match_a.p, n0, (const uint8_t*)c->match_b.p, n1, distance_threshold,
                              lowes_ratio, d_m, d_cnt, desc_bytes <= 61));
    uint64_t cnt = 0;
    AKZ_HIP_TRY(hipMemcpyAsync(&cnt, d_cnt, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
    AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    if (cnt) {
        AKZ_HIP_TRY(hipMemcpyAsync(out, d_m, cnt * sizeof(akz_match), hipMemcpyDeviceToHost, c->stream));
        AKZ_HIP_TRY(hipStreamSynchronize(c->stream));
    }
    *n_out = cnt;
    return AKZ_OK;
}

// Host keypoint logic alone (no GPU): raster-ordered NMS candidates -> keypoints without angle.
// cand: n_cand records {level, idx, v, xp, xm, yp, ym, pad} (32 bytes each, any order).
int akz_host_select_keypoints(uint32_t w, uint32_t h, const akz_config* cfg, const void* cand, uint64_t n_cand,
                              akz_keypoint* out, uint64_t cap, uint64_t* n_out, uint64_t* n_extrema) {
    if (!cfg || (n_cand && !cand) || !n_out) return AKZ_ERR_INVALID_ARG;
    std::vector<LevelPlan> plan;
    AKZ_TRY(build_plan(w, h, *cfg, plan));
    std::vector<Candidate> c((const Candidate*)cand, (const Candidate*)cand + n_cand);
    for (const Candidate& x : c)
        if (x.level >= plan.size() || x.idx >= (uint64_t)plan[x.level].w * plan[x.level].h) {
            set_error("candidate out of range");
            return AKZ_ERR_INVALID_ARG;
        }
    sort_candidates(c, plan);
    std::vector<HostKeypoint> hk;
    uint64_t ne = 0;
    select_keypoints(c.data(), c.size(), plan, *cfg, hk, &ne);
    *n_out = hk.size();
    if (n_extrema) *n_extrema = ne;
    if (out)
        for (size_t i = 0; i < hk.size() && i < cap; ++i)
            out[i] = akz_keypoint{hk[i].x, hk[i].y, hk[i].response, hk[i].size, hk[i].octave, hk[i].class_id, 0.0f, 0};
    return AKZ_OK;
}
int akz_remove_outliers(const akz_keypoint*, uint64_t, const akz_keypoint*, uint64_t, const akz_match*, uint64_t, uint64_t,
                        float, float, akz_match*, uint64_t*);
int akz_match_features(akz_ctx* c, const akz_keypoint* kp0, uint64_t n_kp0, const uint8_t* d0, uint64_t n_d0,
                       const akz_keypoint* kp1, uint64_t n_kp1, const uint8_t* d1, uint64_t n_d1, uint64_t desc_bytes,
                       double lowes_ratio, uint64_t ransac_trials, float ransac_epsilon_inliers, akz_match* out,
                       uint64_t* n_out) {
    if (!n_out) return AKZ_ERR_INVALID_ARG;
    // a match indexes the keypoint lists with descriptor indices (lib.rs:267-274): the reference panics on a set with
    // more descriptors than keypoints as soon as such a match reaches RANSAC; here it is refused up front
    if (n_d0 > n_kp0 || n_d1 > n_kp1) {
        set_error("match_features: a feature set has more descriptors than keypoints");
        return AKZ_ERR_INVALID_ARG;
    }
    std::vector<akz_match> raw((size_t)std::max<uint64_t>(1, n_d0));
    uint64_t n_raw = 0;
    AKZ_TRY(akz_descriptor_match(c, d0, n_d0, d1, n_d1, desc_bytes, 10000, lowes_ratio, raw.data(), &n_raw));  // lib.rs:261-266
    // lib.rs:267-274.  The trials run on the device (akz_fmatrix.hip: the host's model source, same bits) when there is
    // enough of them to pay for a launch and a round trip (~60 us); the samples, the choice of the winner and the final
    // filter stay on the host.  A 4K pair (8 264 matches, 1 000 trials): 1.3-1.4 ms on 16 host threads -> see DESIGN 6.
    TrialsOnDevice on_device;
    if (c && ransac_trials * (n_raw + 4000) >= 400000)
        on_device = [c](const float* x0, const float* y0, const float* x1, const float* y1, uint32_t n, const uint32_t* samples,
                        uint32_t trials, float eps_model, float eps_inlier, float* models, int32_t* inliers) -> int {
            AKZ_TRY(bind(c, true, false));
            const size_t b_pts = (size_t)n * 4 * sizeof(float), b_smp = (size_t)trials * 8 * sizeof(uint32_t);
            const size_t b_mdl = (size_t)trials * 9 * sizeof(float), b_inl = (size_t)trials * sizeof(int32_t);
            auto up = [](size_t v) { return (v + 255) / 256 * 256; };
            const size_t in_bytes = up(b_pts) + up(b_smp), out_bytes = up(b_mdl) + up(b_inl);
            AKZ_TRY(ensure(c, c->ransac_dev, in_bytes + out_bytes));
            AKZ_TRY(ensure_pinned(c, c->ransac_pin, in_bytes + out_bytes));
            char* h = (char*)c->ransac_pin.p;
            char* d = (char*)c->ransac_dev.p;
            std::memcpy(h, x0, (size_t)n * 4); std::memcpy(h + (size_t)n * 4, y0, (size_t)n * 4);
            std::memcpy(h + (size_t)n * 8, x1, (size_t)n * 4); std::memcpy(h + (size_t)n * 12, y1, (size_t)n * 4);
            std::memcpy(h + up(b_pts), samples, b_smp);
            hipStream_t st = c->stream;
            AKZ_HIP_TRY(hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, st));
            launch::ransac_trials(st, (const float*)d, n, (const uint32_t*)(d + up(b_pts)), trials, eps_model, eps_inlier,
                                  (float*)(d + in_bytes), (int32_t*)(d + in_bytes + up(b_mdl)));
            AKZ_HIP_TRY(hipGetLastError());
            AKZ_HIP_TRY(hipMemcpyAsync(h + in_bytes, d + in_bytes, out_bytes, hipMemcpyDeviceToHost, st));
            AKZ_HIP_TRY(hipStreamSynchronize(st));
            std::memcpy(models, h + in_bytes, b_mdl);
            std::memcpy(inliers, h + in_bytes + up(b_mdl), b_inl);
            return AKZ_OK;
        };
    return remove_outliers_impl(kp0, n_kp0, kp1, n_kp1, raw.data(), n_raw, ransac_trials, 0.05f, ransac_epsilon_inliers, out, n_out,
                                on_device);
}

}  // extern "C"
