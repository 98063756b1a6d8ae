// The keypoint half's front end ON THE DEVICE: the extrema candidates into the reference's scan order, who can meet whom in
// the keypoint selection, and the selection itself.
//   * Scan order (scale_space_extrema.rs:32-42 walks the levels in order and each level's pixels in raster order; the
//     detector kernels append their candidates to one list in whatever order their waves finish).  Key = (image, level, flat
//     pixel index), packed into as few bits as the batch needs; rocPRIM's device radix sort of (key, list position) pairs
//     and one gather -- or, for one image's short list, k_sort_rows: ONE launch -- leave the list sorted by image, level and
//     pixel.  Keys are unique, so the result is deterministic.
//   * k_rel_offsets / k_relations: per candidate the earlier candidates of its own / the previous level and those of the next
//     level within `size` of it (pure geometry; the selection's order-dependent questions are asked of these short lists).
//   * k_sel_prepare / k_select / k_sel_pack: the order-dependent selection as a data flow of turns (akz_select.hpp).
// With the host cores of a node shared by 8 ranks, bucketing + sorting were a fifth of the host work of a batch and the
// selection the other four; for a lone image's call the host's selection was the longest piece after the scale space.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>
#include <type_traits>

#include <rocprim/device/device_radix_sort.hpp>

#include "akz_internal.hpp"
#include "akz_gates.hpp"
#include "akz_select.hpp"

namespace akz {
namespace {

struct KeyBits {
    unsigned idx_bits, level_bits, total;
};
__host__ __device__ inline unsigned bits_for(unsigned long long n) {  // bits that hold 0 .. n-1
    unsigned b = 1;
    while (b < 63 && (1ull << b) < n) ++b;
    return b;
}

// list entries beyond the count get the largest key: they sort to the end
__global__ void k_candidate_keys(const Candidate* __restrict__ cand, unsigned cap, const unsigned* __restrict__ d_count,
                                 KeyBits kb, unsigned long long* __restrict__ keys, unsigned* __restrict__ pos) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const unsigned n = min(*d_count, cap);
    unsigned long long k = ~0ull >> (64 - kb.total);
    if (i < n) {
        const Candidate c = cand[i];
        k = ((unsigned long long)c.img << (kb.idx_bits + kb.level_bits)) | ((unsigned long long)c.level << kb.idx_bits) | c.idx;
    }
    keys[i] = k;
    pos[i] = i;
}
__global__ void k_candidate_gather(const Candidate* __restrict__ cand, unsigned cap, const unsigned* __restrict__ d_count,
                                   const unsigned* __restrict__ pos, Candidate* __restrict__ out, unsigned* __restrict__ zero) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*d_count, cap)) return;
    out[i] = cand[pos[i]];
    if (zero) zero[i] = 0u;  // (the reverse-list counters of the device selection, one per list entry)
}

// ---- who can meet whom in the keypoint selection (round 4) ------------------------------------------------------------
// find_scale_space_extrema (scale_space_extrema.rs:43-129) asks, for every candidate in scan order, for the FIRST cache
// entry on its own or the previous level within `size` of it (:57-76), and afterwards drops the entries that have a LATER
// entry of the next level within `size` (:109-129).  Which entries are in the cache at that moment is order-dependent and
// stays on the host; WHO can be within `size` of whom is pure geometry, the same f32 expressions on every candidate pair,
// and is answered here for the whole batch: per candidate the (at most kRel1) candidates of its own level before it and
// of the previous level that lie within its `size` of it, and the (at most kRel2) candidates of the NEXT level within
// `size` of its stored position.  The host's selection then walks these short lists on three small per-image arrays
// (alive, slot, response) instead of querying a 2 MB spatial grid per image: the same answers (tests), a quarter of the
// host time.  Indices are relative to the image's first candidate (u16; 0xffff = none).  A candidate with more neighbours
// than a list holds (coarse levels: `size` spans several level pixels there, and those levels have few candidates) gets
// 0xfffe in the list's first place and the host scans that candidate's partner levels itself; an image with more than
// 65 533 candidates is flagged and takes the grid path.
struct RelLevels {  // per level, as select_keypoints computes them
    float size[kMaxLevels], ratio[kMaxLevels];
    unsigned w[kMaxLevels], h[kMaxLevels];
    unsigned row_base[kMaxLevels + 1];  // where a level's rows begin in an image's row table (h + 1 entries per level)
    unsigned n_levels, n_images;
};
// first list position of every (image, level): offs[img * (L + 1) + level], level == L: the image's end; and of every
// (image, level, row): rows[img * R + row_base[level] + y], y == h: the level's end, R = row_base[L] -- k_relations asks for
// the candidates of a band of rows, and a binary search per question (a dozen dependent loads) was most of its time
__global__ void k_rel_offsets(const Candidate* __restrict__ sorted, const unsigned* __restrict__ d_count, unsigned cap, RelLevels lv,
                              unsigned* __restrict__ offs, unsigned* __restrict__ rows, unsigned* __restrict__ img_flags) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned L = lv.n_levels, R = lv.row_base[L], n_off = lv.n_images * (L + 1);
    if (t >= n_off + lv.n_images * R) return;
    if (t < lv.n_images) img_flags[t] = 0u;  // (only k_relations and k_sel_prepare, which come later, set them)
    unsigned img, level, idx = 0;
    if (t < n_off) {
        img = t / (L + 1);
        level = t - img * (L + 1);
    } else {
        const unsigned u = t - n_off;
        img = u / R;
        const unsigned r = u - img * R;
        level = 0;
        while (level + 1 < L && lv.row_base[level + 1] <= r) ++level;
        idx = (r - lv.row_base[level]) * lv.w[level];
    }
    const unsigned n = min(*d_count, cap);
    unsigned lo = 0, hi = n;  // first entry with (img', level', idx') >= (img, level, idx)
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        const Candidate c = sorted[mid];
        const bool before = c.img < img || (c.img == img && (c.level < level || (c.level == level && c.idx < idx)));
        if (before) lo = mid + 1;
        else hi = mid;
    }
    if (t < n_off) offs[t] = lo;
    else rows[t - n_off] = lo;
}
// One image's list of up to SORT_SMALL_N candidates into scan order in ONE launch of one workgroup (a lone frame's; ten
// launches of keys, rocPRIM's passes and the gather before: 45-55 us of a lone 1080p frame's call, each a few microseconds
// of work): a counting sort over the (level, row) buckets in LDS -- the scan order is level, then row, then column -- and
// an insertion sort of the handful of candidates inside each row.  Keys are unique: the same permutation as the radix sort.
constexpr unsigned SORT_SMALL_N = 8192, SORT_SMALL_ROWS = 16384;
__global__ void __launch_bounds__(1024) k_sort_rows(const Candidate* __restrict__ cand, unsigned cap, const unsigned* __restrict__ d_count,
                                                    RelLevels lv, Candidate* __restrict__ out, unsigned* __restrict__ zero,
                                                    unsigned* __restrict__ offs, unsigned* __restrict__ rows, unsigned* __restrict__ img_flags) {
    __shared__ unsigned s_start[SORT_SMALL_ROWS + 1];  // per bucket: its count, then where it begins
    __shared__ unsigned s_idx[SORT_SMALL_N];
    __shared__ unsigned short s_pos[SORT_SMALL_N];
    __shared__ unsigned s_part[16];
    const unsigned n = min(min(*d_count, cap), SORT_SMALL_N), tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const unsigned R = lv.row_base[lv.n_levels];
    for (unsigned b = tid; b <= R; b += 1024) s_start[b] = 0u;
    __syncthreads();
    constexpr int PER = SORT_SMALL_N / 1024;
    unsigned bucket[PER], slot[PER], idx[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const unsigned i = tid + k * 1024;
        bucket[k] = 0xffffffffu;
        if (i < n) {
            const Candidate c = cand[i];
            const unsigned l = min(c.level, lv.n_levels - 1);
            idx[k] = c.idx;
            bucket[k] = lv.row_base[l] + min(c.idx / lv.w[l], lv.h[l] - 1u);
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (bucket[k] != 0xffffffffu) slot[k] = atomicAdd(&s_start[bucket[k]], 1u);
    __syncthreads();
    // exclusive scan of the counts: a contiguous chunk of buckets per thread
    const unsigned per = (R + 1023) / 1024, b0 = min(R, tid * per), b1 = min(R, b0 + per);
    unsigned mine = 0;
    for (unsigned b = b0; b < b1; ++b) mine += s_start[b];
    unsigned incl = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += v;
    }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    unsigned run = incl - mine;
    for (unsigned w = 0; w < wv; ++w) run += s_part[w];
    for (unsigned b = b0; b < b1; ++b) {
        const unsigned cnt = s_start[b];
        s_start[b] = run;
        run += cnt;
    }
    if (tid == 1023) s_start[R] = n;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k)
        if (bucket[k] != 0xffffffffu) {
            const unsigned at = s_start[bucket[k]] + slot[k];
            s_pos[at] = (unsigned short)(tid + k * 1024);
            s_idx[at] = idx[k];
        }
    __syncthreads();
    for (unsigned b = tid; b < R; b += 1024) {  // inside a row: by column (insertion sort of a handful)
        const unsigned lo = s_start[b], hi = s_start[b + 1];
        for (unsigned i = lo + 1; i < hi; ++i) {
            const unsigned v = s_idx[i];
            const unsigned short pv = s_pos[i];
            unsigned j = i;
            for (; j > lo && s_idx[j - 1] > v; --j) {
                s_idx[j] = s_idx[j - 1];
                s_pos[j] = s_pos[j - 1];
            }
            s_idx[j] = v;
            s_pos[j] = pv;
        }
    }
    __syncthreads();
    for (unsigned i = tid; i < n; i += 1024) {
        out[i] = cand[s_pos[i]];
        if (zero) zero[i] = 0u;
    }
    if (offs) {  // the bucket starts ARE k_rel_offsets' tables (one image): that launch is not needed
        for (unsigned b = tid; b < R; b += 1024) rows[b] = s_start[b];
        if (tid <= lv.n_levels) offs[tid] = tid < lv.n_levels ? s_start[lv.row_base[tid]] : n;
        if (tid == 0) img_flags[0] = 0u;
    }
}
// Longer lists, several images (a lone 4K frame's 26 K candidates, a pair's): the same counting sort over (image, level, row)
// buckets in four launches -- count (a candidate learns its bucket and its place in it), scan of the bucket counts (one
// workgroup, up to SORT_BUCKETS_MAX buckets; leaves the counters zero for the next job and writes k_rel_offsets' tables: the
// bucket starts ARE those tables), place (the columns of a row side by side), rank (a candidate counts the row-mates left of
// it and moves to its final place) -- against eleven (keys, eight rocPRIM passes, gather, offsets).
constexpr unsigned SORT_BUCKETS_MAX = gates::kSortBuckets;
__global__ void __launch_bounds__(256) k_bucket_count(const Candidate* __restrict__ cand, unsigned cap, const unsigned* __restrict__ d_count,
                                                      RelLevels lv, unsigned* __restrict__ cnt, unsigned* __restrict__ bucket_of,
                                                      unsigned* __restrict__ slot_of) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*d_count, cap)) return;
    const Candidate c = cand[i];
    const unsigned l = min(c.level, lv.n_levels - 1), img = min(c.img, lv.n_images - 1);
    const unsigned b = img * lv.row_base[lv.n_levels] + lv.row_base[l] + min(c.idx / lv.w[l], lv.h[l] - 1u);
    bucket_of[i] = b;
    slot_of[i] = atomicAdd(&cnt[b], 1u);
}
// start: nb + 1 entries (the row table of candidate_relations when the job wants its lists, else scratch)
__global__ void __launch_bounds__(1024) k_bucket_scan(unsigned* __restrict__ cnt, unsigned nb, unsigned* __restrict__ start,
                                                      const unsigned* __restrict__ d_count, unsigned cap, RelLevels lv,
                                                      unsigned* __restrict__ offs, unsigned* __restrict__ img_flags) {
    __shared__ unsigned s_part[16];
    __shared__ unsigned s_run;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) s_run = 0u;
    __syncthreads();
    for (unsigned t0 = 0; t0 < nb; t0 += 4096) {  // tiles of 4 096 buckets: four consecutive ones per thread
        const unsigned b0 = t0 + tid * 4;
        unsigned v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = b0 + k < nb ? cnt[b0 + k] : 0u;
            if (b0 + k < nb) cnt[b0 + k] = 0u;  // (the counters are zero again for the next job)
        }
        const unsigned mine = v[0] + v[1] + v[2] + v[3];
        unsigned incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned u = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += u;
        }
        if (lane == 63) s_part[wv] = incl;
        __syncthreads();
        unsigned run = s_run + incl - mine;
        for (unsigned w = 0; w < wv; ++w) run += s_part[w];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (b0 + k < nb) start[b0 + k] = run;
            run += v[k];
        }
        __syncthreads();
        if (tid == 1023) s_run = run;
        __syncthreads();
    }
    const unsigned n = min(*d_count, cap);
    if (tid == 0) start[nb] = n;
    if (offs) {  // candidate_relations' level table and flags (its row table is `start`)
        __syncthreads();  // (this workgroup's own stores to start are visible to it)
        const unsigned L = lv.n_levels, R = lv.row_base[L];
        for (unsigned t = tid; t < lv.n_images * (L + 1); t += 1024) {
            const unsigned img = t / (L + 1), l = t - img * (L + 1);
            offs[t] = l < L ? start[img * R + lv.row_base[l]] : (img + 1 < lv.n_images ? start[(img + 1) * R] : n);
        }
        for (unsigned t = tid; t < lv.n_images; t += 1024) img_flags[t] = 0u;
    }
}
__global__ void __launch_bounds__(256) k_bucket_place(const Candidate* __restrict__ cand, unsigned cap, const unsigned* __restrict__ d_count,
                                                      const unsigned* __restrict__ start, const unsigned* __restrict__ bucket_of,
                                                      const unsigned* __restrict__ slot_of, unsigned* __restrict__ row_keys) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*d_count, cap)) return;
    row_keys[start[bucket_of[i]] + slot_of[i]] = cand[i].idx;  // the row's columns side by side, in arrival order
}
__global__ void __launch_bounds__(256) k_bucket_rank(const Candidate* __restrict__ cand, unsigned cap, const unsigned* __restrict__ d_count,
                                                     const unsigned* __restrict__ start, const unsigned* __restrict__ bucket_of,
                                                     const unsigned* __restrict__ row_keys, Candidate* __restrict__ out,
                                                     unsigned* __restrict__ zero) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*d_count, cap)) return;
    const Candidate c = cand[i];
    const unsigned b = bucket_of[i], lo = start[b], hi = start[b + 1];
    unsigned rank = 0;
    for (unsigned j = lo; j < hi; ++j) rank += row_keys[j] < c.idx ? 1u : 0u;  // (one candidate per pixel: the keys of a row differ)
    out[lo + rank] = c;
    if (zero) zero[lo + rank] = 0u;
}

// One WAVE per candidate (round 6; one thread per candidate before: 20 workgroups for a lone 1080p frame's list, every thread a
// chain of ~15 dependent loads -- 32 us of the call, 55 for a 4K frame's): the lanes test a row band's entries side by side, so a
// scan is one round of loads; matches take their places in list order by a ballot's prefix count, the lists are put together
// in LDS (a wave's LDS operations execute in order) and leave in one store per place.
constexpr int REL_WAVES = 4;  // candidates per workgroup
__global__ void __launch_bounds__(64 * REL_WAVES) k_relations(const Candidate* __restrict__ sorted, const unsigned* __restrict__ d_count, unsigned cap,
                                                              RelLevels lv, const unsigned* __restrict__ offs, const unsigned* __restrict__ rows,
                                                              unsigned short* __restrict__ rel, unsigned* __restrict__ img_flags,
                                                              unsigned* __restrict__ revcnt, unsigned short* __restrict__ rev) {
    __shared__ unsigned short s_out[REL_WAVES][kRel1 + kRel2 + 2];
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned i = blockIdx.x * REL_WAVES + wv;
    if (i >= min(*d_count, cap)) return;  // (wave-uniform; no workgroup barrier below)
    const Candidate c = sorted[i];
    if (c.img >= lv.n_images || c.level >= lv.n_levels) return;
    const unsigned L = lv.n_levels, l = c.level;
    const unsigned* o = offs + (size_t)c.img * (L + 1);
    const unsigned img0 = o[0];
    unsigned short* const mine = s_out[wv];
    if (lane < (unsigned)(kRel1 + kRel2)) mine[lane] = 0xffffu;
    const bool too_many = o[L] - img0 > 65533u;
    bool over1 = false, over2 = false;
    const float size = lv.size[l], size2 = size * size, ratio = lv.ratio[l];
    const unsigned ly = c.idx / lv.w[l], lx = c.idx - ly * lv.w[l];
    const float qx = (float)lx * ratio, qy = (float)ly * ratio;                  // the query point of the first pass (:52-53 of the host code)
    const float px = qx + 0.5f * (ratio - 1.0f), py = qy + 0.5f * (ratio - 1.0f);  // the candidate's stored position
    // partners of level pl in [b, e): rows whose stored y can lie within `reach` of y0 -- a conservative band, the exact
    // test decides.  The band's list positions come from the row table: the loop's bounds do not depend on what it loads.
    const unsigned* rowI = rows + (size_t)c.img * lv.row_base[L];
    auto scan = [&](unsigned pl, unsigned b, unsigned e, float x0, float y0, int first, int room, int& used, bool& overflow) {
        if (b >= e) return;
        const float pr = lv.ratio[pl], off = 0.5f * (pr - 1.0f);
        const unsigned pw = lv.w[pl], ph = lv.h[pl];
        const float reach = size + 1.0f;
        const float ylo = (y0 - reach - off) / pr - 1.0f, yhi = (y0 + reach - off) / pr + 1.0f;
        const unsigned r0 = ylo <= 0.0f ? 0u : (unsigned)ylo;
        if (yhi < 0.0f || r0 >= ph) return;
        const unsigned r1 = min((unsigned)yhi + 1u, ph);  // one past the last row
        const unsigned* rt = rowI + lv.row_base[pl];
        const unsigned jb = max(b, rt[r0]), je = min(e, rt[r1]);
        for (unsigned j0 = jb; j0 < je; j0 += 64u) {  // (all of them wave-uniform)
            const unsigned j = j0 + lane;
            bool hit = false;
            if (j < je) {
                const unsigned pidx = sorted[j].idx;
                const unsigned py_i = pidx / pw, px_i = pidx - py_i * pw;
                const float sx = (float)px_i * pr + off, sy = (float)py_i * pr + off;  // its stored position
                const float dist = (x0 - sx) * (x0 - sx) + (y0 - sy) * (y0 - sy);
                hit = dist <= size2;
            }
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int at_list = used + (int)__popcll(m & ((1ull << lane) - 1ull));  // its place: list order is position order
                if (at_list < room) mine[first + at_list] = (unsigned short)(j - img0);
                if (first == 0 && revcnt) {  // the device selection: j learns that this candidate looks at it (akz_select.hpp)
                    const unsigned at = atomicAdd(&revcnt[j], 1u);
                    if (at < (unsigned)sel::kRev) rev[(size_t)j * sel::kRev + at] = (unsigned short)(i - img0);
                }
            }
            used += (int)__popcll(m);
            if (used > room) overflow = true;
        }
    };
    int used1 = 0, used2 = 0;
    if (l > 0) scan(l - 1, o[l - 1], o[l], qx, qy, 0, kRel1, used1, over1);   // every entry of the previous level comes before this one
    scan(l, o[l], i, qx, qy, 0, kRel1, used1, over1);                          // its own level: the ones before it
    if (l + 1 < L) scan(l + 1, o[l + 1], o[l + 2], px, py, kRel1, kRel2, used2, over2);
    if (lane == 0) {
        if (over1) mine[0] = 0xfffeu;
        if (over2) mine[kRel1] = 0xfffeu;
        if (too_many) img_flags[c.img] = 1u;
    }
    unsigned short* out = rel + (size_t)i * (kRel1 + kRel2);
    if (lane < (unsigned)(kRel1 + kRel2)) out[lane] = mine[lane];
}


// ---- the selection itself (round 5): akz_select.hpp ------------------------------------------------------------------------
// k_sel_prepare (one thread per candidate): everything a candidate's turn needs in one 48-byte row -- its earlier neighbours,
// its rank in the reverse list of each, whether its response beats each neighbour's, its next-level neighbours --, its
// state before the first turn (a candidate without earlier neighbours opens a cache position of its own), and the images
// the device cannot do (a list of earlier neighbours or a reverse list that overflowed): those go to the host's selection.
typedef sel::Row<kRel1, kRel2> SelRow;
static_assert(sizeof(SelRow) == 64 && kRel1 == 12 && kRel2 == 6, "four 16-byte loads; k_select knows where the fields are");
__global__ void __launch_bounds__(256) k_sel_prepare(const Candidate* __restrict__ sorted, const unsigned* __restrict__ d_count, unsigned cap,
                                                     unsigned n_levels, unsigned n_images, const unsigned* __restrict__ offs,
                                                     const unsigned short* __restrict__ rel, const unsigned* __restrict__ revcnt,
                                                     const unsigned short* __restrict__ rev, SelRow* __restrict__ rows,
                                                     unsigned char* __restrict__ state0, unsigned* __restrict__ img_flags) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(*d_count, cap)) return;
    const Candidate c = sorted[i];
    if (c.img >= n_images || c.level >= n_levels) return;
    const unsigned img0 = offs[(size_t)c.img * (n_levels + 1)];
    const unsigned me = i - img0;
    const unsigned short* row = rel + (size_t)i * (kRel1 + kRel2);
    const bool over = row[0] == sel::kListOverflow || revcnt[i] > (unsigned)sel::kRev;  // (an overflowed next-level list: k_select scans that level)
    SelRow out;
    const float mine = fabsf(c.v);
    unsigned wins = 0;
    bool open = true;
#pragma unroll
    for (int j = 0; j < kRel1; ++j) {
        const unsigned short q = row[j];
        unsigned pred = sel::kNone;
        if (!open || q >= sel::kListOverflow) {
            open = false;
        } else {
            const unsigned qi = img0 + q, cnt = min(revcnt[qi], (unsigned)sel::kRev);
            for (unsigned m = 0; m < cnt; ++m) {  // the member of q's reverse list right before this candidate
                const unsigned e = rev[(size_t)qi * sel::kRev + m];
                if (e < me && (pred == sel::kNone || e > pred)) pred = e;
            }
            if (mine > fabsf(sorted[qi].v)) wins |= 1u << j;  // (scale_space_extrema.rs:62: strictly larger replaces)
        }
        out.rel1[j] = open ? q : sel::kNone;
        out.pred[j] = (unsigned short)pred;
    }
    out.wins = (unsigned short)wins;
#pragma unroll
    for (int j = 0; j < kRel2; ++j) out.rel2[j] = row[kRel1 + j];
    {
        const float d_x = 0.5f * (c.xp - c.xm), d_y = 0.5f * (c.yp - c.ym);
        unsigned len = 0;
#pragma unroll
        for (int j = 0; j < kRel1; ++j) len += out.rel1[j] != sel::kNone ? 1u : 0u;
        out.refined = (unsigned short)((fabsf(-d_x) <= 1.0f && fabsf(-d_y) <= 1.0f ? 1u : 0u) | (len << 8));
    }
    rows[i] = out;
    state0[i] = row[0] == sel::kNone ? (unsigned char)1 : (unsigned char)0;  // 1: opens a cache position of its own, no turn to wait for
    if (over) atomicOr(&img_flags[c.img], (row[0] == sel::kListOverflow ? 2u : 0u) | (revcnt[i] > (unsigned)sel::kRev ? 16u : 0u));
}

// k_select: one workgroup per image; ONE 16-bit word per candidate in LDS (undecided / the creator of the cache position it
// sits in / gone: akz_select.hpp).  Thread t owns the candidates t, t + 1024, ... and walks them in index order: it waits for
// the current one's neighbours one after the other -- a neighbour is through when its word and its predecessor's (the
// previous candidate that also looks at it) are decided --, folds each one's contribution in, takes its turn (two stores)
// and moves on.  No barrier separates the turns: whoever sees a predecessor's word decided sees what it stored before (LDS
// operations of a wave execute in order), and the lowest undecided candidate of the image can always go, so every wave
// keeps moving.  The row of the next candidate is fetched while the current one waits.  Then the second pass and the
// refinement's test (a bit per cache position, by its creator), an exclusive scan over the bit words, and every survivor
// writes its keypoint at the rank of its cache position.  One workgroup's loops over tens of candidates per thread are
// bound by the latency of their loads: they fetch for several candidates before they use the first.
constexpr int SEL_NT = 1024;
constexpr unsigned SEL_MAX_CANDS = 65536;      // x 2 bytes = 128 KB of the 160 KB (the lists' 16-bit indices allow 65 533 per image)
constexpr unsigned SEL_MAX_LOOKS = 4u << 20;   // (a turn that never becomes ready cannot happen; if it did, the image goes to the host)
struct SelKp {  // what the host fetches per keypoint: the selection's record, then the orientation sums (k_orientation fills them in)
    sel::KpRec rec;
    OrientOut sums;
};
static_assert(sizeof(SelKp) == 32, "akz_api.cpp fetches 32-byte records");
struct SelOut {
    unsigned* hdr;        // per image, 16 words: keypoints, extrema (before the refinement), status (0: done here), looks of the
                          // slowest thread, 10 ns ticks of the four phases (first states, turns, second pass, output), and what the
                          // host wants with them in ONE copy: the list's length (all images), the image's flags, its contrast factor
    SelKp* recs;          // per image at its first candidate's list position (one image: final)
    KpParam* pars;
    unsigned* total;      // one image: the keypoint count again, for the kernels that follow
    const unsigned* d_count;
    const double* d_k;
};
__device__ __forceinline__ void sel_header_extras(const SelOut& out, unsigned img, unsigned flags) {
    out.hdr[img * 16 + 8] = *out.d_count;
    out.hdr[img * 16 + 9] = flags;
    const double k = out.d_k[img];
    out.hdr[img * 16 + 10] = (unsigned)((unsigned long long)__double_as_longlong(k) & 0xffffffffull);
    out.hdr[img * 16 + 11] = (unsigned)((unsigned long long)__double_as_longlong(k) >> 32);
}
struct SelPacked {  // a row as its sixteen dwords: 0..5 rel1 (two per dword), 6..11 pred, 12 wins | rel2[0], 13..14 rel2[1..4], 15 rel2[5] | refined
    uint4 v[4];
};
__device__ __forceinline__ SelPacked sel_load_row(const SelRow* __restrict__ p) {
    SelPacked r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    r.v[0] = q[0]; r.v[1] = q[1]; r.v[2] = q[2]; r.v[3] = q[3];
    return r;
}
__device__ __forceinline__ void sel_compiler_fence() { __asm__ volatile("" ::: "memory"); }
// The words the waves of k_select exchange while they run are read and written as RELAXED ATOMICS of workgroup scope: the same
// ds_read_u16 / ds_write_b16 as a plain access (no wait is attached to a relaxed operation -- an acquire or release of
// workgroup scope would wait for the global loads in flight, the next row's prefetch, as well), but an access the compiler may
// neither tear, merge with a neighbour, repeat nor hoist out of the loop, and one that is not a data race in the language's
// model.  Order between two of them: sel_compiler_fence() keeps the compiler from moving one across the other, and the LDS
// operations of a wave execute in the order they are issued.
__device__ __forceinline__ unsigned short sel_word_load(const unsigned short* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void sel_word_store(unsigned short* p, unsigned short v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void __launch_bounds__(SEL_NT) k_select(const Candidate* __restrict__ sorted, RelLevels lv, const unsigned* __restrict__ offs,
                                                   const SelRow* __restrict__ rows, const unsigned char* __restrict__ state0,
                                                   const unsigned* __restrict__ img_flags, const unsigned* __restrict__ row_table,
                                                   SelOut out) {
    __shared__ unsigned short s_word[SEL_MAX_CANDS];      // sel::kUndecided / the creator of the candidate's cache position / sel::kGone
    __shared__ unsigned s_bits[SEL_MAX_CANDS / 32];        // per cache position (by its creator): its occupant is a keypoint
    __shared__ unsigned s_pref[SEL_MAX_CANDS / 32];        // keypoints before the word
    __shared__ unsigned s_part[SEL_NT / 64];
    __shared__ unsigned s_extrema, s_abort, s_looks;
    __shared__ float s_ratio[kMaxLevels], s_size[kMaxLevels];  // the level tables by a lane's own index: from LDS, not by a
    __shared__ unsigned s_w[kMaxLevels];                        // per-lane load from the kernel arguments
    const unsigned img = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid < (unsigned)kMaxLevels) {
        s_ratio[tid] = lv.ratio[tid];
        s_size[tid] = lv.size[tid];
        s_w[tid] = lv.w[tid];
    }
    const unsigned L = lv.n_levels;
    const unsigned* o = offs + (size_t)img * (L + 1);
    const unsigned img0 = o[0], n = o[L] - img0;
    unsigned status = img_flags[img];
    if (n > 65533u) status |= 1u;
    if (status != 0u || n == 0u) {
        if (tid == 0) {
            out.hdr[img * 16 + 0] = 0u; out.hdr[img * 16 + 1] = 0u; out.hdr[img * 16 + 2] = status; out.hdr[img * 16 + 3] = 0u;
            sel_header_extras(out, img, img_flags[img]);
            if (out.total) *out.total = 0u;
        }
        return;
    }
    const Candidate* cand = sorted + img0;
    const SelRow* rowI = rows + img0;
    const unsigned long long t0 = wall_clock64();  // (100 MHz; the phases' durations go into the header: akz_debug_select_info)
    if (tid == 0) { s_extrema = 0u; s_abort = 0u; s_looks = 0u; }
    // this thread's candidates: dealt out in turn -- t, t + 1024, ... (a contiguous run per thread, so that a candidate's
    // same-level neighbours are mostly its own thread's, was measured: the threads of a level then start a whole run behind the
    // level before, 1 000 - 1 900 looks on a 4K frame instead of ~100)
    const unsigned per = (n + SEL_NT - 1) / SEL_NT;
    auto cand_of = [&](unsigned k) -> unsigned { return k >= per ? 0xffffffffu : tid + k * SEL_NT; };
    unsigned long long mask = 0ull;  // this thread's undecided candidates: bit k <-> candidate cand_of(k)
    constexpr int U = 8;
    for (unsigned k0 = 0; k0 < per; k0 += U) {
        unsigned char st[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned c = cand_of(k0 + u);
            st[u] = c < n ? state0[img0 + c] : (unsigned char)1;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned c = cand_of(k0 + u);
            if (c < n) {
                s_word[c] = st[u] ? (unsigned short)c : sel::kUndecided;
                if (!st[u]) mask |= 1ull << (k0 + u);
            }
        }
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    auto WORD = [&](unsigned short q) -> unsigned short { return s_word[q]; };
    for (unsigned i = tid; i < SEL_MAX_CANDS / 32; i += SEL_NT) s_bits[i] = 0u;  // (read after the barriers that follow the turns)
    // The loop below runs ~100 times on the slowest thread and a wave pays for its longest path every time: it is kept short.
    // The lists stay packed in their dwords and are shifted down as the neighbours get through; a neighbour's contribution is
    // folded in when it is through (sel::advance), so that the turn itself is three stores.
    SelPacked nxt;
    int b_cur = -1, b_nxt = -1;
    auto fetch = [&](SelPacked& r, int& b) {
        b = -1;
        if (mask) {
            b = __ffsll((long long)mask) - 1;
            mask &= mask - 1ull;
            r = sel_load_row(rowI + cand_of((unsigned)b));
        }
    };
    unsigned wq[6], wp[6], wins = 0, len = 0, hit_q = 0;  // neighbours / their predecessors still waited for (entry 0: the next one)
    sel::Progress prog;
    auto open_row = [&](const SelPacked& r) {
        wq[0] = r.v[0].x; wq[1] = r.v[0].y; wq[2] = r.v[0].z; wq[3] = r.v[0].w; wq[4] = r.v[1].x; wq[5] = r.v[1].y;
        wp[0] = r.v[1].z; wp[1] = r.v[1].w; wp[2] = r.v[2].x; wp[3] = r.v[2].y; wp[4] = r.v[2].z; wp[5] = r.v[2].w;
        wins = r.v[3].x & 0xffffu;
        len = (r.v[3].w >> 24) & 15u;
        // (the row is wanted HERE: without this the compiler lets the lists live in the registers the row was loaded into and
        // waits for every outstanding load -- the next row's prefetch too -- in each look of the loop below)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            AKZ_SEL_KEEP(wq[i]);
            AKZ_SEL_KEEP(wp[i]);
        }
        AKZ_SEL_KEEP(wins);
        AKZ_SEL_KEEP(len);
        sel::start(&prog);
    };
    {
        SelPacked first;
        fetch(first, b_cur);
        if (b_cur >= 0) open_row(first);
    }
    // (the first row has arrived before the loop is entered: the compiler would otherwise wait for "everything outstanding"
    // at the top of every look, i.e. for the prefetch of the next row as well)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    fetch(nxt, b_nxt);
    unsigned looks = 0, waited = 0;
    while (b_cur >= 0) {
        ++looks;
        while ((unsigned)prog.through < len) {
            const unsigned q = wq[0] & 0xffffu, pr = wp[0] & 0xffffu;
            const bool has_pred = pr != sel::kNone;
            // the predecessor's word BEFORE the neighbour's (the predecessor stores "the neighbour is gone" before its own word;
            // LDS operations of a wave execute in the order they are issued)
            sel_compiler_fence();
            const unsigned wpred = sel_word_load(&s_word[has_pred ? pr : 0u]);
            sel_compiler_fence();
            const unsigned wq_now = sel_word_load(&s_word[q]);
            sel_compiler_fence();
            const int before = prog.at;
            if (!sel::advance(&prog, has_pred, (unsigned short)wpred, (unsigned short)wq_now)) break;
            if (prog.at != before) hit_q = q;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                wq[i] = __builtin_amdgcn_alignbit(wq[i + 1], wq[i], 16);
                wp[i] = __builtin_amdgcn_alignbit(wp[i + 1], wp[i], 16);
            }
            wq[5] >>= 16;
            wp[5] >>= 16;
        }
        if ((unsigned)prog.through == len) {
            const unsigned short c = (unsigned short)cand_of((unsigned)b_cur);
            bool replaces;
            const unsigned short mine = sel::turn(prog, c, (unsigned short)wins, &replaces);
            if (replaces) sel_word_store(&s_word[hit_q], sel::kGone);
            sel_compiler_fence();  // (the replaced entry is gone before anyone can see the turn as taken)
            sel_word_store(&s_word[c], mine);
            sel_compiler_fence();
            b_cur = b_nxt;
            if (b_cur >= 0) open_row(nxt);
            fetch(nxt, b_nxt);
            waited = 0;
        } else if ((++waited & 255u) == 0u) {  // (every 256th fruitless look: has the workgroup given up?  A plain LDS read --
            sel_compiler_fence();              // a volatile one is a FLAT load, which waits for the next row's prefetch too)
            if (waited > SEL_MAX_LOOKS || __hip_atomic_load(&s_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
                __hip_atomic_store(&s_abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
            }
        }
    }
    atomicMax(&s_looks, looks);
    __syncthreads();
    const unsigned long long t2 = wall_clock64();
    if (s_abort) status = 4u;
    if (status != 0u) {
        if (tid == 0) {
            out.hdr[img * 16 + 0] = 0u; out.hdr[img * 16 + 1] = 0u; out.hdr[img * 16 + 2] = status; out.hdr[img * 16 + 3] = s_looks;
            sel_header_extras(out, img, img_flags[img]);
            if (out.total) *out.total = 0u;
        }
        return;
    }
    // second pass (:109-129) and the refinement's test (:141-178) on this thread's candidates
    unsigned long long surv = 0ull;
    unsigned my_extrema = 0;
    constexpr int U2 = 4, U3 = 4;
    for (unsigned k0 = 0; k0 < per; k0 += U3) {
        uint4 tail[U3];  // the row's last 16 bytes: wins, the next-level neighbours, the refinement's flag
#pragma unroll
        for (int u = 0; u < U3; ++u) {
            const unsigned c = cand_of(k0 + u);
            tail[u] = make_uint4(0u, 0u, 0u, 0u);
            if (c < n && sel::alive(s_word[c])) tail[u] = reinterpret_cast<const uint4*>(rowI + c)[3];
        }
#pragma unroll
        for (int u = 0; u < U3; ++u) {
            const unsigned c = cand_of(k0 + u);
            if (c >= n || !sel::alive(s_word[c])) continue;
            union { uint4 v; unsigned short h[8]; } t;
            t.v = tail[u];
            const unsigned short* r2 = &t.h[1];  // (bytes 50 .. 61 of the row)
            const unsigned short mine = s_word[c];
            bool repeated;
            if (r2[0] == sel::kListOverflow) {
                // more next-level neighbours than the list holds: that level's candidates in the band of rows that can lie within
                // `size` of this one's stored position, with the selection's own expressions (as the host does for such a candidate)
                repeated = false;
                const Candidate cd = cand[c];
                const unsigned l = cd.level, w = s_w[l];
                if (l + 1 < L) {
                    const float ratio = s_ratio[l], size = s_size[l], size2 = size * size;
                    const unsigned ly = cd.idx / w, lx = cd.idx - ly * w;
                    const float px = (float)lx * ratio + 0.5f * (ratio - 1.0f), py = (float)ly * ratio + 0.5f * (ratio - 1.0f);
                    const float pr = s_ratio[l + 1], off = 0.5f * (pr - 1.0f), reach = size + 1.0f;
                    const unsigned pw = s_w[l + 1], ph = lv.h[l + 1];
                    const float ylo = (py - reach - off) / pr - 1.0f, yhi = (py + reach - off) / pr + 1.0f;
                    const unsigned r0 = ylo <= 0.0f ? 0u : (unsigned)ylo;
                    if (yhi >= 0.0f && r0 < ph) {
                        const unsigned* rt = row_table + (size_t)img * lv.row_base[L] + lv.row_base[l + 1];  // (k_rel_offsets' row table)
                        const unsigned jb = rt[r0], je = rt[min((unsigned)yhi + 1u, ph)];
                        for (unsigned j = jb; j < je && !repeated; ++j) {
                            const Candidate p = sorted[j];
                            const unsigned q = j - img0;
                            if (!sel::alive(s_word[q]) || s_word[q] < mine) continue;
                            const unsigned qy = p.idx / pw, qx = p.idx - qy * pw;
                            const float sx = (float)qx * pr + off, sy = (float)qy * pr + off;
                            const float dist = (px - sx) * (px - sx) + (py - sy) * (py - sy);
                            repeated = dist <= size2;
                        }
                    }
                }
            } else {
                repeated = sel::repeated_later<kRel2>(mine, r2, WORD);
            }
            if (repeated) continue;
            ++my_extrema;
            if (t.h[7] & 1u) surv |= 1ull << (k0 + u);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_extrema += __shfl_xor(my_extrema, off, 64);
    if (lane == 0 && my_extrema) atomicAdd(&s_extrema, my_extrema);
    __syncthreads();
    const unsigned long long t3 = wall_clock64();
    for (unsigned long long m = surv; m; m &= m - 1ull) {
        const unsigned org = s_word[cand_of((unsigned)(__ffsll((long long)m) - 1))];
        atomicOr(&s_bits[org >> 5], 1u << (org & 31u));
    }
    __syncthreads();
    // keypoints in cache order = in the order of the positions' creators: an exclusive scan over the bit words
    constexpr unsigned WPT = SEL_MAX_CANDS / 32 / SEL_NT;  // bit words per thread
    unsigned cnt[WPT], mine_n = 0;
#pragma unroll
    for (unsigned k = 0; k < WPT; ++k) {
        cnt[k] = __popc(s_bits[tid * WPT + k]);
        mine_n += cnt[k];
    }
    unsigned incl = mine_n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned v = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += v;
    }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    unsigned before = 0, total = 0;
    for (unsigned w = 0; w < SEL_NT / 64; ++w) {
        const unsigned v = s_part[w];
        if (w < wv) before += v;
        total += v;
    }
    unsigned run = before + incl - mine_n;
#pragma unroll
    for (unsigned k = 0; k < WPT; ++k) {
        s_pref[tid * WPT + k] = run;
        run += cnt[k];
    }
    __syncthreads();
    // every survivor's keypoint at the rank of its cache position
    for (unsigned long long m = surv; m;) {
        unsigned cs[U2];
        Candidate cds[U2];
        int nb = 0;
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            cs[u] = 0xffffffffu;
            if (m) {
                cs[u] = cand_of((unsigned)(__ffsll((long long)m) - 1));
                m &= m - 1ull;
                cds[u] = cand[cs[u]];
                ++nb;
            }
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            if (cs[u] == 0xffffffffu) continue;
            const Candidate cd = cds[u];
            const unsigned org = s_word[cs[u]];
            const unsigned at = s_pref[org >> 5] + __popc(s_bits[org >> 5] & ((1u << (org & 31u)) - 1u));
            const unsigned l = cd.level, w = s_w[l];
            const unsigned ly = cd.idx / w, lx = cd.idx - ly * w;
            const float ratio = s_ratio[l];
            sel::KpRec rec;
            rec.level = l;
            (void)sel::refine(lx, ly, cd.v, cd.xp, cd.xm, cd.yp, cd.ym, ratio, &rec);
            KpParam p;
            p.xf = rec.x / ratio;
            p.yf = rec.y / ratio;
            p.scale = roundf(0.5f * s_size[l] / ratio);
            p.level = l;
            p.img = img;
            p._pad[0] = p._pad[1] = p._pad[2] = 0u;
            out.recs[img0 + at].rec = rec;
            out.pars[img0 + at] = p;
        }
        (void)nb;
    }
    if (tid == 0) {
        out.hdr[img * 16 + 0] = total; out.hdr[img * 16 + 1] = s_extrema; out.hdr[img * 16 + 2] = 0u; out.hdr[img * 16 + 3] = s_looks;
        out.hdr[img * 16 + 4] = (unsigned)(t1 - t0); out.hdr[img * 16 + 5] = (unsigned)(t2 - t1); out.hdr[img * 16 + 6] = (unsigned)(t3 - t2);
        out.hdr[img * 16 + 7] = (unsigned)(wall_clock64() - t3);
        sel_header_extras(out, img, img_flags[img]);
        if (out.total) *out.total = total;
    }
}

// several images: the per-image runs into one list in image order (the keypoint kernels and the descriptor rows index it)
__global__ void __launch_bounds__(256) k_sel_pack(const unsigned* __restrict__ offs, unsigned n_levels, unsigned n_images,
                                                  const unsigned* __restrict__ hdr, const SelKp* __restrict__ recs_in,
                                                  const KpParam* __restrict__ pars_in, SelKp* __restrict__ recs, KpParam* __restrict__ pars,
                                                  unsigned* __restrict__ total) {
    __shared__ unsigned s_red[256];
    const unsigned img = blockIdx.x, tid = threadIdx.x;
    unsigned part = 0;
    for (unsigned j = tid; j < img; j += 256) part += hdr[j * 16];
    s_red[tid] = part;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)tid < off) s_red[tid] += s_red[tid + off];
        __syncthreads();
    }
    const unsigned base = s_red[0], cnt = hdr[img * 16], img0 = offs[(size_t)img * (n_levels + 1)];
    for (unsigned i = blockIdx.y * 256 + tid; i < cnt; i += gridDim.y * 256) {
        recs[base + i].rec = recs_in[img0 + i].rec;
        pars[base + i] = pars_in[img0 + i];
    }
    if (img + 1 == n_images && blockIdx.y == 0 && tid == 0) *total = base + cnt;
}

}  // namespace

namespace launch {

static size_t up256(size_t b) { return (b + 255) / 256 * 256; }
// rel scratch: level offsets | image flags | neighbour lists | row table
static uint32_t rows_per_image(const uint32_t* level_h, uint32_t n_levels) {
    uint32_t r = 0;
    for (uint32_t l = 0; l < n_levels; ++l) r += level_h[l] + 1;
    return r;
}
size_t candidate_relations_bytes(uint32_t cap, const uint32_t* level_h, uint32_t n_levels, uint32_t n_images) {
    return up256((size_t)n_images * (n_levels + 1) * 4) + up256((size_t)n_images * 4) + up256((size_t)cap * (kRel1 + kRel2) * sizeof(uint16_t)) +
           up256(((size_t)n_images * rows_per_image(level_h, n_levels) + 1) * 4) + 256;  // (+ 1: the bucket sort's end entry)
}
static void fill_levels(RelLevels& lv, const float* size, const float* ratio, const uint32_t* level_w, const uint32_t* level_h, uint32_t n_levels,
                        uint32_t n_images) {
    std::memset(&lv, 0, sizeof(lv));
    uint32_t base = 0;
    for (uint32_t l = 0; l < n_levels && l < (uint32_t)kMaxLevels; ++l) {
        lv.size[l] = size[l];
        lv.ratio[l] = ratio[l];
        lv.w[l] = level_w[l];
        lv.h[l] = level_h[l];
        lv.row_base[l] = base;
        base += level_h[l] + 1;
    }
    lv.row_base[std::min<uint32_t>(n_levels, kMaxLevels)] = base;
    lv.n_levels = n_levels;
    lv.n_images = n_images;
}
// scratch of the device selection: reverse-list counters | reverse lists | rows | first states | headers | total |
// per-image keypoint runs (several images)
struct SelLayout {
    size_t revcnt, rev, rows, state0, hdr, total, recs_tmp, pars_tmp, end;
};
static SelLayout sel_layout(uint32_t cap, uint32_t n_images) {
    SelLayout l;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t b = at; at += up256(bytes); return b; };
    l.revcnt = take((size_t)cap * 4);
    l.rev = take((size_t)cap * sel::kRev * 2);
    l.rows = take((size_t)cap * sizeof(SelRow));
    l.state0 = take(cap);
    l.hdr = take((size_t)n_images * 64);
    l.total = take(4);
    l.recs_tmp = take(n_images > 1 ? (size_t)cap * sizeof(SelKp) : 0);
    l.pars_tmp = take(n_images > 1 ? (size_t)cap * sizeof(KpParam) : 0);
    l.end = at;
    return l;
}
size_t select_device_bytes(uint32_t cap, uint32_t n_images) { return sel_layout(cap, n_images).end + 256; }
uint32_t* select_device_revcnt(void* sel_scratch, uint32_t cap, uint32_t n_images) {
    return (uint32_t*)((char*)sel_scratch + sel_layout(cap, n_images).revcnt);
}
void select_device(hipStream_t s, const Candidate* d_sorted, uint32_t cap, const uint32_t* d_count, const float* size, const float* ratio,
                   const uint32_t* level_w, const uint32_t* level_h, uint32_t n_levels, uint32_t n_images, const void* rel_scratch,
                   void* sel_scratch, const double* d_k, void* d_recs, KpParam* d_pars, uint32_t** d_hdr_out, uint32_t** d_total_out) {
    const SelLayout l = sel_layout(cap, n_images);
    char* p = (char*)sel_scratch;
    const size_t offs_b = up256((size_t)n_images * (n_levels + 1) * 4), flags_b = up256((size_t)n_images * 4);
    const unsigned* offs = (const unsigned*)rel_scratch;
    unsigned* flags = (unsigned*)((char*)rel_scratch + offs_b);
    const unsigned short* rel = (const unsigned short*)((const char*)rel_scratch + offs_b + flags_b);
    const unsigned* row_table = (const unsigned*)((const char*)rel_scratch + offs_b + flags_b + up256((size_t)cap * (kRel1 + kRel2) * sizeof(uint16_t)));
    RelLevels lv;
    fill_levels(lv, size, ratio, level_w, level_h, n_levels, n_images);
    hipLaunchKernelGGL(k_sel_prepare, dim3((cap + 255) / 256), dim3(256), 0, s, d_sorted, d_count, cap, n_levels, n_images, offs, rel,
                       (const unsigned*)(p + l.revcnt), (const unsigned short*)(p + l.rev), (SelRow*)(p + l.rows),
                       (unsigned char*)(p + l.state0), flags);
    SelOut out;
    out.hdr = (unsigned*)(p + l.hdr);
    unsigned* total = (unsigned*)(p + l.total);
    const bool one = n_images == 1;
    out.recs = one ? (SelKp*)d_recs : (SelKp*)(p + l.recs_tmp);
    out.d_count = d_count;
    out.d_k = d_k;
    out.pars = one ? d_pars : (KpParam*)(p + l.pars_tmp);
    out.total = one ? total : nullptr;
    hipLaunchKernelGGL(k_select, dim3(n_images), dim3(SEL_NT), 0, s, d_sorted, lv, offs, (const SelRow*)(p + l.rows),
                       (const unsigned char*)(p + l.state0), (const unsigned*)flags, row_table, out);
    if (!one)
        hipLaunchKernelGGL(k_sel_pack, dim3(n_images, 4), dim3(256), 0, s, offs, n_levels, n_images, (const unsigned*)out.hdr,
                           (const SelKp*)out.recs, (const KpParam*)out.pars, (SelKp*)d_recs, d_pars, total);
    *d_hdr_out = out.hdr;
    *d_total_out = total;
}
// d_rel_out / d_flags_out: where the lists and the per-image overflow flags are inside `scratch`
void candidate_relations(hipStream_t s, const Candidate* d_sorted, uint32_t cap, const uint32_t* d_count, const float* size, const float* ratio,
                         const uint32_t* level_w, const uint32_t* level_h, uint32_t n_levels, uint32_t n_images, void* scratch,
                         uint16_t** d_rel_out, uint32_t** d_flags_out, void* sel_scratch, bool tables_ready) {
    const size_t offs_b = up256((size_t)n_images * (n_levels + 1) * 4), flags_b = up256((size_t)n_images * 4),
                 rel_b = up256((size_t)cap * (kRel1 + kRel2) * sizeof(uint16_t));
    unsigned* offs = (unsigned*)scratch;
    unsigned* flags = (unsigned*)((char*)scratch + offs_b);
    unsigned short* rel = (unsigned short*)((char*)scratch + offs_b + flags_b);
    unsigned* rows = (unsigned*)((char*)scratch + offs_b + flags_b + rel_b);
    *d_rel_out = rel;
    *d_flags_out = flags;
    RelLevels lv;
    fill_levels(lv, size, ratio, level_w, level_h, n_levels, n_images);
    const unsigned no = n_images * (n_levels + 1) + n_images * lv.row_base[std::min<uint32_t>(n_levels, kMaxLevels)];
    if (!tables_ready)  // (sort_candidates_rows leaves them behind)
        hipLaunchKernelGGL(k_rel_offsets, dim3((no + 255) / 256), dim3(256), 0, s, d_sorted, d_count, cap, lv, offs, rows, flags);
    unsigned* revcnt = nullptr;
    unsigned short* rev = nullptr;
    if (sel_scratch) {  // (the counters were zeroed by the sort's gather)
        const SelLayout l = sel_layout(cap, n_images);
        revcnt = (unsigned*)((char*)sel_scratch + l.revcnt);
        rev = (unsigned short*)((char*)sel_scratch + l.rev);
    }
    hipLaunchKernelGGL(k_relations, dim3((cap + REL_WAVES - 1) / REL_WAVES), dim3(64 * REL_WAVES), 0, s, d_sorted, d_count, cap, lv, offs, rows, rel, flags,
                       revcnt, rev);
}

// scratch layout: keys in | keys out | positions in | positions out | rocPRIM's temporary storage
static size_t sort_temp(uint32_t cap, unsigned end_bit) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (unsigned*)nullptr,
                                    (unsigned*)nullptr, cap, 0, end_bit, nullptr);
    return bytes;
}
static KeyBits key_bits(uint64_t max_px, uint32_t n_levels, uint32_t n_images) {
    KeyBits kb;
    kb.idx_bits = bits_for(max_px);
    kb.level_bits = bits_for(n_levels);
    kb.total = kb.idx_bits + kb.level_bits + bits_for((unsigned long long)n_images + 1);  // + 1: the all-ones padding key
    return kb;
}
uint32_t sort_small_capacity() { return SORT_SMALL_N; }
bool sort_candidates_rows(hipStream_t s, const Candidate* d_cand, uint32_t cap, const uint32_t* d_count, const uint32_t* level_w,
                          const uint32_t* level_h, uint32_t n_levels, Candidate* d_sorted, uint32_t* d_zero, void* rel_scratch) {
    if (cap > SORT_SMALL_N || n_levels == 0 || n_levels > (uint32_t)kMaxLevels || rows_per_image(level_h, n_levels) > SORT_SMALL_ROWS) return false;
    RelLevels lv;
    std::vector<float> none(n_levels, 0.0f);
    fill_levels(lv, none.data(), none.data(), level_w, level_h, n_levels, 1);
    unsigned *offs = nullptr, *rows = nullptr, *flags = nullptr;
    if (rel_scratch) {  // (candidate_relations' layout for one image)
        const size_t offs_b = up256((size_t)(n_levels + 1) * 4), flags_b = up256(4), rel_b = up256((size_t)cap * (kRel1 + kRel2) * sizeof(uint16_t));
        offs = (unsigned*)rel_scratch;
        flags = (unsigned*)((char*)rel_scratch + offs_b);
        rows = (unsigned*)((char*)rel_scratch + offs_b + flags_b + rel_b);
    }
    hipLaunchKernelGGL(k_sort_rows, dim3(1), dim3(1024), 0, s, d_cand, cap, d_count, lv, d_sorted, d_zero, offs, rows, flags);
    return true;
}
// scratch of the bucket sort: counters (zero between jobs) | starts | bucket of a candidate | its place | the rows' keys
size_t sort_candidates_buckets_scratch(uint32_t cap) {
    return up256((size_t)(SORT_BUCKETS_MAX + 1) * 4) * 2 + up256((size_t)cap * 4) * 3 + 256;
}
bool sort_candidates_buckets(hipStream_t s, const Candidate* d_cand, uint32_t cap, const uint32_t* d_count, const uint32_t* level_w,
                             const uint32_t* level_h, uint32_t n_levels, uint32_t n_images, void* scratch, bool scratch_is_new,
                             Candidate* d_sorted, uint32_t* d_zero, void* rel_scratch) {
    if (n_levels == 0 || n_levels > (uint32_t)kMaxLevels || cap == 0) return false;
    const uint64_t nb64 = (uint64_t)n_images * rows_per_image(level_h, n_levels);
    if (nb64 > SORT_BUCKETS_MAX) return false;
    const unsigned nb = (unsigned)nb64;
    RelLevels lv;
    std::vector<float> none(n_levels, 0.0f);
    fill_levels(lv, none.data(), none.data(), level_w, level_h, n_levels, n_images);
    char* p = (char*)scratch;
    const size_t cb = up256((size_t)(SORT_BUCKETS_MAX + 1) * 4), lb = up256((size_t)cap * 4);
    unsigned* cnt = (unsigned*)p;
    unsigned* start = (unsigned*)(p + cb);
    unsigned* bucket_of = (unsigned*)(p + 2 * cb);
    unsigned* slot_of = (unsigned*)(p + 2 * cb + lb);
    unsigned* row_keys = (unsigned*)(p + 2 * cb + 2 * lb);
    if (scratch_is_new) (void)hipMemsetAsync(cnt, 0, cb, s);  // (afterwards k_bucket_scan leaves them zero)
    unsigned *offs = nullptr, *flags = nullptr;
    if (rel_scratch) {  // (candidate_relations' layout: its row table holds the bucket starts)
        const size_t offs_b = up256((size_t)n_images * (n_levels + 1) * 4), flags_b = up256((size_t)n_images * 4),
                     rel_b = up256((size_t)cap * (kRel1 + kRel2) * sizeof(uint16_t));
        offs = (unsigned*)rel_scratch;
        flags = (unsigned*)((char*)rel_scratch + offs_b);
        start = (unsigned*)((char*)rel_scratch + offs_b + flags_b + rel_b);
    }
    const unsigned gb = (cap + 255) / 256;
    hipLaunchKernelGGL(k_bucket_count, dim3(gb), dim3(256), 0, s, d_cand, cap, d_count, lv, cnt, bucket_of, slot_of);
    hipLaunchKernelGGL(k_bucket_scan, dim3(1), dim3(1024), 0, s, cnt, nb, start, d_count, cap, lv, offs, flags);
    hipLaunchKernelGGL(k_bucket_place, dim3(gb), dim3(256), 0, s, d_cand, cap, d_count, (const unsigned*)start, (const unsigned*)bucket_of,
                       (const unsigned*)slot_of, row_keys);
    hipLaunchKernelGGL(k_bucket_rank, dim3(gb), dim3(256), 0, s, d_cand, cap, d_count, (const unsigned*)start, (const unsigned*)bucket_of,
                       (const unsigned*)row_keys, d_sorted, d_zero);
    return true;
}
size_t sort_candidates_scratch(uint32_t cap, uint64_t max_px, uint32_t n_levels, uint32_t n_images) {
    const KeyBits kb = key_bits(max_px, n_levels, n_images);
    const size_t a = ((size_t)cap * 8 + 255) / 256 * 256, b = ((size_t)cap * 4 + 255) / 256 * 256;
    return 2 * a + 2 * b + sort_temp(cap, kb.total) + 256;
}
bool sort_candidates_device(hipStream_t s, const Candidate* d_cand, uint32_t cap, const uint32_t* d_count, uint64_t max_px,
                            uint32_t n_levels, uint32_t n_images, void* scratch, Candidate* d_sorted, uint32_t* d_zero) {
    if (cap == 0) return true;
    const KeyBits kb = key_bits(max_px, n_levels, n_images);
    if (kb.total > 63) return false;
    const size_t a = ((size_t)cap * 8 + 255) / 256 * 256, b = ((size_t)cap * 4 + 255) / 256 * 256;
    char* p = (char*)scratch;
    unsigned long long* k_in = (unsigned long long*)p;
    unsigned long long* k_out = (unsigned long long*)(p + a);
    unsigned* p_in = (unsigned*)(p + 2 * a);
    unsigned* p_out = (unsigned*)(p + 2 * a + b);
    void* tmp = p + 2 * a + 2 * b;
    size_t tmp_bytes = sort_temp(cap, kb.total);
    const unsigned nb = (cap + 255) / 256;
    hipLaunchKernelGGL(k_candidate_keys, dim3(nb), dim3(256), 0, s, d_cand, cap, d_count, kb, k_in, p_in);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, p_in, p_out, cap, 0, kb.total, s) != hipSuccess) return false;
    hipLaunchKernelGGL(k_candidate_gather, dim3(nb), dim3(256), 0, s, d_cand, cap, d_count, p_out, d_sorted, d_zero);
    return true;
}

}  // namespace launch
}  // namespace akz
