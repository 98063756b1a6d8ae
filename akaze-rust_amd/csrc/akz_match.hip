// Brute-force Hamming matcher on the matrix cores (gfx950).
//
// ops::feature_matching::descriptor_match (akaze/src/ops/feature_matching.rs:23-94) scans, for every query
// descriptor, all train descriptors and keeps the smallest and second smallest Hamming distance (strict '<'
// updates, :41-49: the lowest index wins among equal minima).  With the descriptor bits unpacked to int8 0/1,
//     hamming(a, b) = |a| + |b| - 2 <a, b>      (a: train row, b: query row),
// and the inner products of all (train, query) pairs are one integer GEMM with K = 512: exact integer
// arithmetic, so distances, indices and therefore the match list are identical to the popcount scan
// (k_match in akz_kernels.hip, which stays the path for small sets), at a multiple of its rate — the popcount
// kernel is bound by the vector ALU (16 xor + 16 bit-count-accumulate per pair), this one by
// v_mfma_i32_32x32x32_i8.
//
// Layout.  k_unpack_bits writes a set as int8 [rows padded][512] (bit b of byte t -> column 8t + b; any fixed
// order works, both operands use the same one; the free columns 488..511 carry the train rows' bit counts, see
// there).  k_match_mfma: 1024 threads = 16 waves, four per SIMD; a wave owns 32 queries as the B operand for all 16
// K-steps (64 VGPRs, loaded once); the workgroup walks a chunk of the train set in LDS tiles of 128 rows (four
// 32-row MFMA tiles per barrier) that all waves share (double-buffered; row pitch 528 B so that the 16-byte
// operand reads of a 16-lane group fall into different banks; the next tile arrives in 32-row parts through one
// register stage).  A/B operand of lane l (r = l & 31, h = l >> 5) at K-step s: bytes 32 s + 16 h .. + 15 of
// row r — the same function of (l, s) for both operands, which is all the dot product needs.  The 32 x 32 result
// has its column (query) on the lane and rows (i & 3) + 8 (i >> 2) + 4 h, i = 0..15, in the registers
// (cdna_hip_programming.md, C/D layout): ascending train index inside a lane, so the reference's update rule
// applies directly; the two lanes of a column are merged at the end of the chunk with the order-free form of the
// rule ((distance, index) lexicographic minimum; second = min of the others), and the chunks by k_match_compact.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdlib>

#include "akz_internal.hpp"

namespace akz {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int KB = 512;            // int8 columns per descriptor (64 bytes x 8 bits)
// 32-column query blocks per wave.  With 2, every A operand read from LDS feeds two MFMAs (half the LDS traffic per
// product) but the 128 resident operand registers leave room for only 8 waves per workgroup: measured 25-35 % SLOWER at
// 11 K and 90 K rows (profiles/r02_match_variants.txt) -- the loop is bound by the MFMA pipe and its latency hiding,
// not by LDS bandwidth -- so the default stays 1 (16 waves, four per SIMD).
#ifndef AKZ_MM_NB
#define AKZ_MM_NB 1
#endif
constexpr int MM_NB = AKZ_MM_NB;
constexpr int MM_NT = 1024 / MM_NB;  // threads per workgroup
constexpr int MM_QW = 32 * MM_NB;  // queries per wave; their B operands stay resident (64 VGPRs per block)
constexpr int MM_QB = (MM_NT / 64) * MM_QW;  // queries per workgroup (512)
#ifndef AKZ_MM_SUB
#define AKZ_MM_SUB 4
#endif
#ifndef AKZ_MM_AHEAD
#define AKZ_MM_AHEAD 3
#endif
constexpr int MM_SUB = AKZ_MM_SUB;  // 32-row MFMA tiles per LDS tile, i.e. per barrier (1: 2.91, 2: 2.97, 4: 3.07 T pairs/s)
constexpr int MM_TR = 32 * MM_SUB;  // train rows per LDS tile
constexpr int MM_PITCH = KB + 16;  // LDS row pitch in bytes

// One wave per descriptor row: lane t expands byte t into 8 int8 values; the row's bit count by wave reduction.
// A descriptor has 61 bytes (486 bits used, feature_matching.rs works on those bytes), so columns 488..511 are free.
// They fold the train row's bit count into the product: a TRAIN row gets its bits as 0/1 and its count split over
// columns 488..491 (each <= 127), a QUERY row gets its bits as 0/2 and -1 in those four columns, so that
//     <train row, query row> = 2 <a, b> - |a|     and     hamming = |b| - that,
// i.e. the kernel's epilogue is a maximum over accumulators instead of 16 additions of row counts.
// tiles (optional): for images of SEVERAL sets laid out one after the other, each padded to whole LDS tiles — entry
// t = {first source row, number of valid rows} of padded rows t * MM_TR .. + MM_TR - 1 (n is ignored then).
// fp4 (the form k_match_fp4 reads): 4 bits per column, 256 bytes per row -- bit 1 -> +1.0, bit 0 -> -1.0 (e2m1 codes
// 0x2 / 0xA) in the 488 columns of the 61 descriptor bytes, 0 in the other 24, the same for queries and train rows:
//     <a', b'> = 488 - 2 hamming(a, b),
// so no bit counts are needed at all (and every sum is a small integer, exact in the f32 accumulators).
__device__ __forceinline__ void unpack_row(unsigned row, unsigned lane, const uint8_t* __restrict__ d, unsigned n, unsigned n_pad, bool query,
                                           uint8_t* __restrict__ out, unsigned* __restrict__ pop,
                                           unsigned* __restrict__ bound, unsigned threshold, unsigned n_bound,
                                           const uint2* __restrict__ tiles, bool fp4) {
    if (row >= n_pad) return;
    unsigned src = row;
    bool live = row < n;
    if (tiles) {
        const uint2 e = tiles[row / MM_TR];
        const unsigned local = row % MM_TR;
        live = local < e.y;
        src = e.x + local;
    }
    const unsigned byte = (live && lane < 61u) ? d[(size_t)src * 64 + lane] : 0u;
    unsigned c = __popc(byte);
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (fp4) {
        unsigned v4 = 0u;
        if (live && lane < 61u) {
#pragma unroll
            for (int b = 0; b < 8; ++b) v4 |= (((byte >> b) & 1u) ? 0x2u : 0xAu) << (4 * b);
        }
        *reinterpret_cast<unsigned*>(out + (size_t)row * (KB / 2) + 4 * lane) = v4;
        if (lane == 0) {
            pop[row] = c;
            if (bound)
                for (unsigned k = 0; k < n_bound; ++k) bound[(size_t)k * n_pad + row] = threshold;
        }
        return;
    }
    uint2 v;
    v.x = (byte & 1u) | ((byte & 2u) << 7) | ((byte & 4u) << 14) | ((byte & 8u) << 21);
    v.y = ((byte >> 4) & 1u) | (((byte >> 4) & 2u) << 7) | (((byte >> 4) & 4u) << 14) | (((byte >> 4) & 8u) << 21);
    if (query) {
        v.x <<= 1;
        v.y <<= 1;
        if (lane == 61u) v.x = 0xffffffffu;  // -1 in columns 488..491
    } else if (lane == 61u) {
        const unsigned c0 = min(c, 127u), c1 = min(c - c0, 127u), c2 = min(c - c0 - c1, 127u), c3 = c - c0 - c1 - c2;  // c <= 488
        v.x = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
    }
    *reinterpret_cast<uint2*>(out + (size_t)row * KB + 8 * lane) = v;
    if (lane == 0) {
        pop[row] = c;
        // queries: the shared pruning bounds of k_match_mfma (one per train set) start at the threshold
        if (bound)
            for (unsigned k = 0; k < n_bound; ++k) bound[(size_t)k * n_pad + row] = threshold;
    }
}

// The FP4 form with FOUR rows per wave: 16 lanes per row, a lane expands one u32 of the row (four descriptor bytes) into 32
// e2m1 codes (one 16-byte store) -- a quarter of the waves of unpack_row and word-wide loads (the pair call's unpack launch
// 9.7 -> 5 us).  Same bytes out as unpack_row(fp4): bit 1 -> 0x2 (+1.0), bit 0 -> 0xA (-1.0) for bytes 0..60, zero beyond.
__device__ __forceinline__ unsigned fp4_codes_of_byte(unsigned b) {  // bit i of b -> nibble i
    unsigned x = (b | (b << 12)) & 0x000F000Fu;
    x = (x | (x << 6)) & 0x03030303u;
    x = (x | (x << 3)) & 0x11111111u;
    return 0xAAAAAAAAu ^ (x << 3);
}
__device__ __forceinline__ void unpack_rows_fp4(unsigned row, unsigned sub, const uint8_t* __restrict__ d, unsigned n, unsigned n_pad,
                                                uint8_t* __restrict__ out, unsigned* __restrict__ pop, unsigned* __restrict__ bound,
                                                unsigned threshold, unsigned n_bound, const uint2* __restrict__ tiles) {
    if (row >= n_pad) return;  // (whole 16-lane groups leave together)
    unsigned src = row;
    bool live = row < n;
    if (tiles) {
        const uint2 e = tiles[row / MM_TR];
        const unsigned local = row % MM_TR;
        live = local < e.y;
        src = e.x + local;
    }
    unsigned word = live ? reinterpret_cast<const unsigned*>(d + (size_t)src * 64)[sub] : 0u;
    if (sub == 15u) word &= 0xffu;  // bytes 61..63 are padding: never compared (feature_matching.rs works on the 61 bytes)
    unsigned c = __popc(word);
    for (int o = 8; o > 0; o >>= 1) c += __shfl_xor(c, o, 16);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (live) {
        v.x = fp4_codes_of_byte(word & 0xffu);
        if (sub < 15u) {
            v.y = fp4_codes_of_byte((word >> 8) & 0xffu);
            v.z = fp4_codes_of_byte((word >> 16) & 0xffu);
            v.w = fp4_codes_of_byte(word >> 24);
        }
    }
    *reinterpret_cast<uint4*>(out + (size_t)row * (KB / 2) + 16 * sub) = v;
    if (sub == 0) {
        pop[row] = c;
        if (bound)
            for (unsigned k = 0; k < n_bound; ++k) bound[(size_t)k * n_pad + row] = threshold;
    }
}
__global__ void __launch_bounds__(256) k_unpack_bits_fp4(const uint8_t* __restrict__ d, unsigned n, unsigned n_pad, uint8_t* __restrict__ out,
                                                         unsigned* __restrict__ pop, unsigned* __restrict__ bound, unsigned threshold,
                                                         unsigned n_bound, const uint2* __restrict__ tiles) {
    unpack_rows_fp4(blockIdx.x * 16 + (threadIdx.x >> 4), threadIdx.x & 15u, d, n, n_pad, out, pop, bound, threshold, n_bound, tiles);
}
__global__ void __launch_bounds__(256) k_unpack_pair_fp4(const uint8_t* __restrict__ dq, unsigned nq, unsigned q_pad, uint8_t* __restrict__ outq,
                                                         unsigned* __restrict__ popq, unsigned* __restrict__ bound, unsigned threshold,
                                                         const uint8_t* __restrict__ dt, unsigned nt, unsigned t_pad, uint8_t* __restrict__ outt,
                                                         unsigned* __restrict__ popt) {
    const unsigned row = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15u;  // (q_pad is a multiple of 16: a group is all query or all train)
    if (row < q_pad) unpack_rows_fp4(row, sub, dq, nq, q_pad, outq, popq, bound, threshold, 1u, nullptr);
    else unpack_rows_fp4(row - q_pad, sub, dt, nt, t_pad, outt, popt, nullptr, 0u, 0u, nullptr);
}
__global__ void __launch_bounds__(256) k_unpack_bits(const uint8_t* __restrict__ d, unsigned n, unsigned n_pad, bool query,
                                                     uint8_t* __restrict__ out, unsigned* __restrict__ pop,
                                                     unsigned* __restrict__ bound, unsigned threshold, unsigned n_bound,
                                                     const uint2* __restrict__ tiles, bool fp4) {
    unpack_row(blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63u, d, n, n_pad, query, out, pop, bound, threshold, n_bound, tiles, fp4);
}
// a pair call's two sets in one launch: rows [0, q_pad) are the query set, the rest the train set
__global__ void __launch_bounds__(256) k_unpack_pair(const uint8_t* __restrict__ dq, unsigned nq, unsigned q_pad, uint8_t* __restrict__ outq,
                                                     unsigned* __restrict__ popq, unsigned* __restrict__ bound, unsigned threshold,
                                                     const uint8_t* __restrict__ dt, unsigned nt, unsigned t_pad, uint8_t* __restrict__ outt,
                                                     unsigned* __restrict__ popt, bool fp4) {
    const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (row < q_pad) unpack_row(row, lane, dq, nq, q_pad, true, outq, popq, bound, threshold, 1u, nullptr, fp4);
    else unpack_row(row - q_pad, lane, dt, nt, t_pad, false, outt, popt, nullptr, 0u, 0u, nullptr, fp4);
}

// blockIdx.x: 512 queries; blockIdx.y: a chunk of `chunk_tiles` train tiles.  q8 / t8: unpacked sets, rows padded
// to a multiple of MM_QB / MM_TR with zero rows (k_unpack_bits, query / train form); qpop: bit counts of the queries.  Writes (min, second, argmin) of every
// live query over the chunk to out[chunk * n0 + query].
// A chunk of a multi-set launch (k_match_mfma with a table): LDS tiles [t_begin, t_end) of the padded train image
// belong to one set that starts at padded row row0 and has n_rows valid rows; records go to out[record * n0 + query]
// with row indices relative to the set, bounds live at bound[bound_off + query].
struct MatchChunk {
    unsigned t_begin, t_end, row0, n_rows, bound_off, record;
};

__global__ void __launch_bounds__(MM_NT) k_match_mfma(const uint8_t* __restrict__ q8, const unsigned* __restrict__ qpop,
                                                      unsigned n0, const uint8_t* __restrict__ t8,
                                                      unsigned n1, unsigned chunk_tiles,
                                                      unsigned threshold, unsigned* __restrict__ bound,
                                                      MatchRec* __restrict__ out, const MatchChunk* __restrict__ table) {
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[2][MM_TR * MM_PITCH];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned r = lane & 31u, h = lane >> 5;
    const unsigned q_first = blockIdx.x * MM_QB + wave * MM_QW;  // this wave's 64 queries: block b = queries q_first + 32 b ..

    // B operands: 16 K-steps of each of this wave's query blocks, resident for the whole chunk
    v4i bq[MM_NB][16];
#pragma unroll
    for (int b = 0; b < MM_NB; ++b) {
        const uint8_t* row = q8 + (size_t)(q_first + 32 * b + r) * KB + 16 * h;
#pragma unroll
        for (int s = 0; s < 16; ++s) bq[b][s] = *reinterpret_cast<const v4i*>(row + 32 * s);
    }
    // pin the operands here: otherwise the waits for these loads are placed at their first use inside the tile loop
#pragma unroll
    for (int b = 0; b < MM_NB; ++b)
#pragma unroll
        for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(bq[b][s]));
    unsigned pq[MM_NB], min_d[MM_NB], second[MM_NB], min_j[MM_NB];
    // A chunk starts without knowing anything about its queries, and while `second` is still large nearly every tile
    // contains a row that beats it for one of the wave's queries (the slow path below then runs for the whole wave).
    // All workgroups that scan for a query therefore share an upper bound of its final second-best distance: bound[q]
    // only ever receives some chunk's current `second`, which is never below the final value, so rows with a distance
    // ABOVE the bound cannot appear in the final (min, second, argmin) — rows AT the bound still take the exact path
    // (they can decide a tie) — and skipping them leaves the merged result identical.  The bound is re-read every
    // fourth tile past the L1 (agent-scope atomic load) and the value is used four tiles later: never waited for.
    unsigned limit[MM_NB];   // min(second, bound[q] + 1): a tile whose best distance is >= limit changes nothing that matters
    unsigned b_seen[MM_NB];  // bound[q] as last read
#pragma unroll
    for (int b = 0; b < MM_NB; ++b) {
        pq[b] = qpop[q_first + 32 * b + r];
        min_d[b] = second[b] = limit[b] = b_seen[b] = threshold;
        min_j[b] = 0u;
    }

    unsigned t_begin, t_end, row0 = 0u, record = blockIdx.y;
    if (table) {  // one train set per chunk
        const MatchChunk ck = table[blockIdx.y];
        t_begin = ck.t_begin; t_end = ck.t_end; row0 = ck.row0; n1 = ck.n_rows; record = ck.record;
        bound += ck.bound_off;
    } else {
        const unsigned tiles_total = (n1 + MM_TR - 1) / MM_TR;
        t_begin = blockIdx.y * chunk_tiles;
        t_end = min(tiles_total, t_begin + chunk_tiles);
    }
    // Staging: the next LDS tile is fetched one 32-row part at a time (32 rows x 32 sixteen-byte pieces = 1024 pieces,
    // two per thread): part p is requested before the MFMA chain of sub-tile p of the current tile and handed to the
    // other LDS buffer after that sub-tile's epilogue, so one register stage serves MM_SUB parts per barrier.
    constexpr int PIECES = 32 * KB / 16 / MM_NT;
    static_assert(PIECES * MM_NT * 16 == 32 * KB, "whole pieces per thread and 32-row part");
    unsigned st_src[PIECES], st_dst[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const unsigned idx = tid + (unsigned)p * MM_NT;
        st_src[p] = (idx >> 5) * KB + (idx & 31u) * 16u;
        st_dst[p] = (idx >> 5) * MM_PITCH + (idx & 31u) * 16u;
    }
    uint4 stage[PIECES];
    auto fetch = [&](unsigned tile, int part) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            stage[p] = *reinterpret_cast<const uint4*>(t8 + ((size_t)tile * MM_TR + 32u * part) * KB + st_src[p]);
    };
    auto commit = [&](int buf, int part) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) *reinterpret_cast<uint4*>(&s_tile[buf][32 * part * MM_PITCH + st_dst[p]]) = stage[p];
    };
    if (t_begin < t_end) {
#pragma unroll
        for (int part = 0; part < MM_SUB; ++part) {
            fetch(t_begin, part);
            commit(0, part);
        }
    }
    __syncthreads();
    for (unsigned tile = t_begin; tile < t_end; ++tile) {
        const int buf = (int)((tile - t_begin) & 1u);
        const bool more = tile + 1 < t_end;
        const bool partial = (tile + 1) * MM_TR - row0 > n1;  // uniform: only the last tile of the set
#ifndef AKZ_MM_REFRESH
#define AKZ_MM_REFRESH 1  // tiles between two reads of the shared bound (power of two)
#endif
        if (((tile - t_begin) & (AKZ_MM_REFRESH - 1u)) == 0u) {  // use the value requested last time, request the next one
#pragma unroll
            for (int b = 0; b < MM_NB; ++b) {
                limit[b] = min(limit[b], min(second[b], b_seen[b] < 0xffffffffu ? b_seen[b] + 1u : b_seen[b]));
                b_seen[b] = __hip_atomic_load(bound + q_first + 32 * b + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // (Sharing the bound between the two lanes of a column directly -- max of their minima bounds the second-best
        // too -- was tried: any cross-lane instruction here costs 38 spilled registers at the 128 of 16 waves, and the
        // lanes already meet through `bound`, which is now read every tile.)
#pragma unroll
        for (int sub = 0; sub < MM_SUB; ++sub) {
            if (more) fetch(tile + 1, sub);  // in flight under the MFMA chain below
            v16i acc[MM_NB];
#pragma unroll
            for (int b = 0; b < MM_NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = 0;
            const uint8_t* arow = &s_tile[buf][(32 * sub + r) * MM_PITCH + 16 * h];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const v4i a = *reinterpret_cast<const v4i*>(arow + 32 * s);
#pragma unroll
                for (int b = 0; b < MM_NB; ++b) acc[b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[b][s], acc[b], 0, 0, 0);
            }
            // schedule: keep AKZ_MM_AHEAD operand reads in flight ahead of the MFMAs that consume them (an LDS read
            // takes ~100 cycles, an MFMA 32; left alone the scheduler issues read, wait, MFMA, read, wait, ...)
#if AKZ_MM_AHEAD > 0
            __builtin_amdgcn_sched_group_barrier(0x100, AKZ_MM_AHEAD, 0);
#pragma unroll
            for (int s = 0; s < 16 - AKZ_MM_AHEAD; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, MM_NB, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, AKZ_MM_AHEAD * MM_NB, 0);
#endif
            // distances of this lane's 16 train rows of the sub-tile (ascending index) to its queries
            const unsigned j0 = tile * MM_TR - row0 + 32 * sub + 4 * h;  // row index inside the set
#pragma unroll
            for (int b = 0; b < MM_NB; ++b) {
                // acc[i] = 2 <a, b> - |a| (the row counts ride in the product), so hamming = |b| - acc[i]: the smallest
                // distance of the 16 rows is pq minus the largest accumulator
                int top = acc[b][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) top = max(top, acc[b][i]);
                #ifdef AKZ_MM_NOSLOW
                if (partial) {
#else
                if (partial || (int)pq[b] - top < (int)limit[b]) {  // rare: see `limit`
#endif
                    // exact update from the tile's two smallest distances and the first row of the smallest: keys
                    // (acc << 4 | 15 - i) order by accumulator, then by ascending row; top two keys by max3 / med3
                    int key[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const unsigned row = (unsigned)((i & 3) + 8 * (i >> 2));
                        key[i] = (partial && j0 + row >= n1) ? INT_MIN : (acc[b][i] << 4) + (15 - i);  // padding rows never match
                    }
                    int M = INT_MIN, S = INT_MIN;
#pragma unroll
                    for (int g = 0; g < 5; ++g) {  // five groups of three keys, then the sixteenth
                        const int ka = key[3 * g], kb = key[3 * g + 1], kc = key[3 * g + 2];
                        const int gm = max(ka, max(kb, kc)), gs = max(min(ka, kb), min(max(ka, kb), kc));  // largest, median
                        S = max(min(M, gm), max(S, gs));
                        M = max(M, gm);
                    }
                    S = max(S, min(M, key[15]));
                    M = max(M, key[15]);
                    const int i1 = 15 - (M & 15);
                    const unsigned tb = M == INT_MIN ? 0xffffffffu : (unsigned)((int)pq[b] - (M >> 4));
                    const unsigned ts = S == INT_MIN ? 0xffffffffu : (unsigned)((int)pq[b] - (S >> 4));
                    const unsigned before = second[b];
                    if (tb < min_d[b]) {  // rows of a lane arrive in ascending order: the sequential rule (:41-49) folded per tile
                        second[b] = min(min_d[b], ts);
                        min_d[b] = tb;
                        min_j[b] = j0 + (unsigned)((i1 & 3) + 8 * (i1 >> 2));
                    } else {
                        second[b] = min(second[b], tb);
                    }
                    if (second[b] < before) {
                        atomicMin(bound + q_first + 32 * b + r, second[b]);
                        limit[b] = min(limit[b], second[b]);
                    }
                }
            }
            if (more) commit(buf ^ 1, sub);
        }
        __syncthreads();
    }
    // the two lanes of a column hold disjoint row sets: order-free merge, then one record per live query
#pragma unroll
    for (int b = 0; b < MM_NB; ++b) {
        const unsigned o_min = __shfl_xor(min_d[b], 32, 64), o_sec = __shfl_xor(second[b], 32, 64);
        const unsigned o_j = __shfl_xor(min_j[b], 32, 64);
        unsigned m = min_d[b], s2 = second[b], j = min_j[b];
        if (o_min < m || (o_min == m && o_min < threshold && o_j < j)) {  // the other lane holds the winner
            s2 = min(o_sec, m);
            m = o_min;
            j = o_j;
        } else {
            s2 = min(s2, o_min);
        }
        const unsigned q = q_first + 32 * b + r;
        if (h == 0 && q < n0) {
            MatchRec rec;
            rec.min_d = m; rec.second_d = s2; rec.min_j = j; rec._pad = 0;
            out[(size_t)record * n0 + q] = rec;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same scan on the block-scaled FP4 matrix instruction (gfx950: v_mfma_scale_f32_32x32x64_f8f6f4, twice the K per
// instruction and per operand byte of the int8 form at the same issue cost).  Operands are the +-1 images of
// k_unpack_bits(fp4): e2m1 holds +-1 exactly, the block scales are 1 (E8M0 127), and a dot product of 488 terms of +-1 is
// an integer of magnitude <= 488: the f32 accumulators are exact.  hamming = (488 - dot) / 2.  Structure, tie rule and
// output are those of k_match_mfma; a train row is 256 bytes, a wave's query operands 32 registers.  What differs since
// round 4: the exact top-2 update costs what the case needs (see `The exact update` in the kernel), the second-distance
// bound shared between workgroups is off (AKZ_MM4_BOUND), the one-direction form takes two tiles per barrier (AKZ_MM4_STEP),
// and a staged half tile goes to LDS before the chain's accumulators are looked at.
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int KB4 = KB / 2;           // bytes per unpacked row
constexpr int MM_PITCH4 = KB4 + 16;   // LDS row pitch: 16-byte operand reads of 16 consecutive rows fall into different banks
constexpr int kBits = 488;            // columns that carry +-1 (61 bytes)
constexpr float kPadAcc = -1.0e30f;   // accumulator given to the padding rows of a partial tile (a real one is >= -488)
constexpr float kPadKey = -1.0e5f;    // ... and what tells their keys 16 acc + (15 - i) from real ones (>= -7808)

// NB: 32-query blocks per wave.  1 (used): 16 waves of 32 queries (1024 threads); 2: 8 waves of 64 queries (512 threads),
// every train operand read from LDS feeding two matrix instructions -- measured 30 % SLOWER at 90 K rows (2.36 against
// 1.80 ms, profiles/r03_match_variants.txt), as for the int8 form: the loop needs its 16 waves.
#ifndef AKZ_MM4_NB
#define AKZ_MM4_NB 1
#endif
#ifndef AKZ_MM4_NT
#define AKZ_MM4_NT 1024
#endif
// AKZ_MM4_STEP: tiles per barrier of the one-direction form (two LDS buffers of STEP tiles each).
// AKZ_MM4_BOUND = 1: the workgroups that scan for a query share an upper bound of its second distance (bound[q], as in
// k_match_mfma; pushed once per tile).  With the exact update at its round-4 cost that pays only for very large sets
// (89 816 x 89 816: 1551 -> 1495 us); a 4K pair (61 against 64 us), the multi-set launch of the all-pairs step (478 against
// 491 us) and the step itself do better without the atomics and the re-reads, so it is off (profiles/r04_match_mutual.txt).
#ifndef AKZ_MM4_STEP
#define AKZ_MM4_STEP 2
#endif
// AKZ_MM4_RING: tiles staged ahead in the both-direction form, one barrier per that many tiles (2: 7.32 -> 7.17 ms for the
// all-pairs step of 16 x 4K 5x5 frames; the limits a lane compares with are then two tiles old instead of one).
#ifndef AKZ_MM4_RING
#define AKZ_MM4_RING 2
#endif
#ifndef AKZ_MM4_BOUND
#define AKZ_MM4_BOUND 0
#endif
constexpr int MM4_QB = (AKZ_MM4_NT / 64) * 32 * AKZ_MM4_NB;  // queries per workgroup of the FP4 kernel
// BOTH DIRECTIONS of a block in one pass (COLS; akz_descriptor_match_sets_mutual_device): hamming is symmetric, so the
// distances of the queries to a train set are also the distances of that set's rows to the queries, and the match list
// of the opposite direction (feature_matching.rs:23-94 with the two sets exchanged) needs, for every TRAIN row, the two
// smallest distances over the queries and the lowest query index among the minima.  A train row's candidates sit in
// different lanes, waves and workgroups, so its state lives in memory: cbest[row] = (distance << 32 | query), csecond[row],
// both indexed by the row of the padded train image; an element enters with
//     old = atomicMin(cbest, mine);  atomicMin(csecond, max(old, mine) >> 32)
// (the second smallest distance is the smallest among the losers of the first exchange; the packed minimum keeps the lowest
// query among equal distances).  Elements ABOVE the row's current csecond cannot be one of its final two and are skipped:
// the rows' bounds travel with the tile -- staged one tile ahead into LDS as accumulator limits 488 - 2 csecond (stale
// values are only looser) -- and every lane compares its 16 accumulators with the limits of its 16 rows (four 16-byte LDS
// reads that a whole half-wave shares); only a true candidate leaves the straight-line path.  The state starts from the
// exact result over the first col_q0 queries (a separate small launch of this kernel with the roles exchanged,
// launch::match_cols_seed): from a bound of thousands of samples on, a train row sees a handful of candidates in all.
// (Measured on the way, 16 x 4K 5x5 frames, 120 blocks: gate on the loosest bound of a sub-tile + per-candidate bound loads
// 58 ms; bulk bound loads, chunks rotated against the query blocks, 2048 seed rows 18.5 ms -- of which 7 ms the bound loads
// and 1.6 ms the atomics; limits in LDS 13.1 ms; compares OR-ed on the scalar unit 12.5 ms; candidates found by scalar
// masks, losers settled after the next commit 8.8 ms; lead images on two streams 7.3 ms.  A third of the sub-tiles hold a
// candidate -- 3.3 per train row and lead image, as 2 ln(rows / seed rows) says -- and what one costs is the time its wave
// keeps the workgroup waiting at the tile's barrier: profiles/r04_match_mutual.txt.)
template <int NB, int NT, bool COLS = false>
__global__ void __launch_bounds__(NT) k_match_fp4(const uint8_t* __restrict__ q4, unsigned n0, const uint8_t* __restrict__ t4,
                                                  unsigned n1, unsigned chunk_tiles, unsigned threshold,
                                                  unsigned* __restrict__ bound, MatchRec* __restrict__ out,
                                                  const MatchChunk* __restrict__ table,
                                                  unsigned long long* __restrict__ cbest = nullptr,
                                                  unsigned* __restrict__ csecond = nullptr, unsigned col_q0 = 0) {
    constexpr int QB = (NT / 64) * 32 * NB;
    // (12 waves -- NT = 768, one or two query blocks per wave -- measured in round 4: 2.68 / 2.44 ms against 1.80 at 90 K rows,
    //  profiles/r04_match_mutual.txt: the loop wants its 16 waves; that form is not maintained)
    static_assert(MM_SUB == 4 && (NB == 1 || NB == 2) && (NT == 512 || NT == 1024), "staging below: two 64-row parts per 128-row tile");
    static_assert(!COLS || NB == 1, "the opposite direction's pending exchange: one query per lane");
    // Two buffers of STEP tiles each: one barrier per STEP tiles (4 STEP sub-tiles).  The waves of a workgroup meet at that
    // barrier, so a step takes what its slowest wave takes, and what differs between the waves -- exact updates, candidates of
    // the opposite direction -- averages out over more sub-tiles (round 4, profiles/r04_match_mutual.txt).
    // (two tiles per step: 89 816 x 89 816 1596 -> 1557 us, multi-set launch 481 -> 470 us; with the opposite direction one
    //  tile -- its limits are a step old when they are used, and two tiles cost it 11 spilled registers: 8.8 -> 9.6 ms)
    constexpr int STEP = (COLS || NT < 1024) ? 1 : AKZ_MM4_STEP, SUBS = MM_SUB * STEP;  // (two 512-thread workgroups per CU: one tile each)
    // AHEAD (the both-direction form): steps staged ahead in a ring of 2 AHEAD buffers, one barrier per AHEAD steps -- the same
    // halving of the barriers without the doubled loop body (AKZ_MM4_RING)
    constexpr int AHEAD = (COLS && STEP == 1 && NT == 1024) ? AKZ_MM4_RING : 1, NBUF = 2 * AHEAD;
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[NBUF][STEP * MM_TR * MM_PITCH4];
    __shared__ __attribute__((aligned(16))) float s_lim[NBUF][COLS ? STEP * MM_TR : 4];  // COLS: accumulator limits of the rows
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned r = lane & 31u, h = lane >> 5;
    const unsigned q_first = blockIdx.x * QB + wave * 32u * NB;
    auto op = [](v4i x) { return v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0}; };  // FP4 operands occupy the first four registers

    v4i bq[NB][8];  // B operands: the wave's queries, eight K-steps of 64 columns, resident for the whole chunk
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const uint8_t* row = q4 + (size_t)(q_first + 32 * b + r) * KB4 + 16 * h;
#pragma unroll
        for (int s = 0; s < 8; ++s) bq[b][s] = *reinterpret_cast<const v4i*>(row + 32 * s);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(bq[b][s]));
    unsigned min_d[NB], second[NB], limit[NB], b_seen[NB], pushed[NB], min_j[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        min_d[b] = second[b] = limit[b] = b_seen[b] = pushed[b] = threshold;
        min_j[b] = 0u;
    }
    (void)pushed;
    (void)bound;  // (AKZ_MM4_BOUND 0)

    unsigned t_begin, t_end, row0 = 0u, record = blockIdx.y;
    if (table) {
        // COLS: workgroups that are dispatched together (consecutive query blocks) take DIFFERENT chunks, so that a train row
        // meets the query blocks one after the other and its bound tightens between the visits
        const unsigned chunk = COLS ? (blockIdx.y + blockIdx.x * 7u) % gridDim.y : blockIdx.y;
        const MatchChunk ck = table[chunk];
        t_begin = ck.t_begin; t_end = ck.t_end; row0 = ck.row0; n1 = ck.n_rows; record = ck.record;
        bound += ck.bound_off;
    } else {
        const unsigned tiles_total = (n1 + MM_TR - 1) / MM_TR;
        t_begin = blockIdx.y * chunk_tiles;
        t_end = min(tiles_total, t_begin + chunk_tiles);
    }
    // staging: the next step's tiles arrive in 64-row parts (1024 sixteen-byte pieces each) through one register stage:
    // part p requested before chain 2 p, handed to the other LDS buffer after chain 2 p + 1 has been issued
    constexpr int PIECES = (1024 + NT - 1) / NT;
    constexpr bool EXACT = PIECES * NT == 1024;  // (768 threads: the second piece exists for the first 256 only)
    unsigned st_src[PIECES], st_dst[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
        const unsigned idx = min(tid + (unsigned)p * NT, 1023u);  // (a surplus thread repeats piece 1023: same bytes, same place)
        st_src[p] = (idx >> 4) * KB4 + (idx & 15u) * 16u;
        st_dst[p] = (idx >> 4) * MM_PITCH4 + (idx & 15u) * 16u;
    }
    (void)EXACT;
    uint4 stage[PIECES];
    unsigned long long lim_lo = 0, lim_hi = 0;  // COLS: csecond of rows 4 tid .. 4 tid + 3 of the tile being staged (tid < 32)
    // COLS: the lane's last exchange with a train row's cbest whose loser has not been entered in csecond yet
    unsigned long long pend_old = 0ull;
    unsigned pend_d = 0u, pend_row = 0xffffffffu;
#ifdef AKZ_MM_COUNT
    __shared__ unsigned s_count[4];
    if (tid < 4) s_count[tid] = 0;
#endif
    auto settle = [&]() {
        if constexpr (COLS) {
            if (pend_row != 0xffffffffu) {
#ifdef AKZ_MM_COUNT
                if (pend_old > (((unsigned long long)pend_d << 32) | (q_first + r))) atomicAdd(&s_count[2], 1u);
#endif
                const unsigned long long mine = ((unsigned long long)pend_d << 32) | (q_first + r);
                const unsigned loser = (unsigned)((pend_old > mine ? pend_old : mine) >> 32);
                if (loser < threshold) atomicMin(csecond + pend_row, loser);
                pend_row = 0xffffffffu;
            }
        }
    };
    auto fetch = [&](unsigned tile, int part) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p)
            stage[p] = *reinterpret_cast<const uint4*>(t4 + ((size_t)tile * MM_TR + 64u * part) * KB4 + st_src[p]);
        if constexpr (COLS) {
            if ((part & 1) == 0 && tid < 32u) {  // (the bounds of a tile's 128 rows come with its first part)
                const unsigned long long* src =
                    reinterpret_cast<const unsigned long long*>(csecond + (size_t)tile * MM_TR + 64u * part + 4u * tid);
                lim_lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lim_hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    auto commit = [&](int buf, int part, unsigned tile) {
#pragma unroll
        for (int p = 0; p < PIECES; ++p) *reinterpret_cast<uint4*>(&s_tile[buf][64 * part * MM_PITCH4 + st_dst[p]]) = stage[p];
        if constexpr (COLS) {
            if ((part & 1) == 0 && tid < 32u) {
                const unsigned sec[4] = {(unsigned)lim_lo, (unsigned)(lim_lo >> 32), (unsigned)lim_hi, (unsigned)(lim_hi >> 32)};
                float4 lim;
                float* lp = &lim.x;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned row = tile * MM_TR + 64u * part - row0 + 4u * tid + (unsigned)k;  // row of the set: padding rows never take part
                    lp[k] = row < n1 ? (float)(kBits - 2 * (int)min(sec[k], 1000u)) : 1e9f;
                }
                *reinterpret_cast<float4*>(&s_lim[buf][64u * part + 4u * tid]) = lim;
            }
        }
    };
#pragma unroll
    for (int ah = 0; ah < AHEAD; ++ah) {
        const unsigned t0 = t_begin + (unsigned)(ah * STEP);
        if (t0 < t_end) {
#pragma unroll
            for (int part = 0; part < 2 * STEP; ++part) {
                if (part < 2 * (int)min((unsigned)STEP, t_end - t0)) {
                    fetch(t0, part);
                    commit(ah, part, t0);
                }
            }
        }
    }
    __syncthreads();
    for (unsigned tile = t_begin; tile < t_end; tile += STEP) {
        const unsigned step = (tile - t_begin) / STEP;
        const int buf = (int)(step % NBUF), buf_to = (int)((step + AHEAD) % NBUF);
        const unsigned to = tile + AHEAD * STEP;                                                 // the step staged during this one
        const unsigned here = min((unsigned)STEP, t_end - tile);                                 // tiles of this step
        const unsigned next = to < t_end ? min((unsigned)STEP, t_end - to) : 0u;                  // ... and of the one being staged
#if AKZ_MM4_BOUND > 0
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            limit[b] = min(limit[b], min(second[b], b_seen[b] < 0xffffffffu ? b_seen[b] + 1u : b_seen[b]));
            if (second[b] < min(pushed[b], b_seen[b])) {  // once per tile, and only what the others do not know yet
                atomicMin(bound + q_first + 32 * b + r, second[b]);
                pushed[b] = second[b];
            }
            b_seen[b] = __hip_atomic_load(bound + q_first + 32 * b + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
#pragma unroll
        for (int sub = 0; sub < SUBS; ++sub) {
            if (sub >= MM_SUB && (unsigned)sub >= MM_SUB * here) break;  // (uniform: the last step of a chunk may be short)
            const bool more = (unsigned)(sub >> 1) < 2u * next;          // part sub / 2 of the next step exists
            const bool partial = (tile + (sub / MM_SUB) + 1) * MM_TR - row0 > n1;  // uniform: only the last tile of the set
            if (more && (sub & 1) == 0) fetch(to, sub >> 1);  // in flight under the MFMA chains below
            v16f acc[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = 0.0f;
            const uint8_t* arow = &s_tile[buf][(32 * sub + r) * MM_PITCH4 + 16 * h];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const v4i a = *reinterpret_cast<const v4i*>(arow + 32 * s);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    acc[b] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(op(a), op(bq[b][s]), acc[b], 4, 4, 0, 127, 0, 127);
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int s = 0; s < 8 - 3; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NB, 0);
            // The half tile requested before the previous chain goes to LDS while this chain runs, BEFORE its accumulators
            // are looked at: the wait for the loads is a wait for everything the wave has in flight, the atomics of the
            // opposite direction included (vmcnt counts them for 600..3000 cycles), and here those are a chain or two old.
            // Their answers are in by then too: settled without another wait.
            if ((sub & 1) == 1) {
                if (more) commit(buf_to, sub >> 1, to);
                settle();
            }
            const unsigned j0 = tile * MM_TR - row0 + 32 * sub + 4 * h;  // row index inside the set
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (partial) {  // uniform, the last tile of a set: padding rows never match
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (j0 + (unsigned)((i & 3) + 8 * (i >> 2)) >= n1) acc[b][i] = kPadAcc;
                }
                float topf = acc[b][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) topf = fmaxf(topf, acc[b][i]);
                // the smallest distance of the lane's 16 rows (the accumulators are 488 - 2 d: the half is exact; as a float, so
                // that 16 padding rows stay above any threshold)
                const float bestf = ((float)kBits - topf) * 0.5f;
#ifdef AKZ_MM_COLS_NOCMP  // measurement: the pass without the opposite direction's compares
                if constexpr (false) {
#else
                if constexpr (COLS) {
#endif
                    // the limits of the lane's 16 rows (four groups of four consecutive rows, shared by the half-wave)
                    const float* lp = &s_lim[buf][32 * sub + 4 * h];
                    float lim[16];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const float4 v = *reinterpret_cast<const float4*>(lp + 8 * g4);
                        lim[4 * g4] = v.x; lim[4 * g4 + 1] = v.y; lim[4 * g4 + 2] = v.z; lim[4 * g4 + 3] = v.w;
                    }
                    // One compare per accumulator; the 16 lane masks stay on the scalar unit.  A sub-tile with a candidate
                    // (30..60 % of them: 1024 pairs at 2 / samples-so-far each) is then scanned mask by mask with scalar
                    // branches, and only the accumulator that holds one is looked at by its lanes.  (Round 4: as a per-lane
                    // loop over the 16 accumulators -- compare, branch, execution mask each -- the scan was 2.5 ms of the
                    // 10.3 ms all-pairs step; `hit = hit || ...` had the compiler build a 16-bit mask per lane out of ~50
                    // vector instructions on every sub-tile.)
                    unsigned long long hit[16], any = 0ull;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        hit[i] = __ballot(acc[b][i] >= lim[i]);
                        any |= hit[i];
                    }
                    const unsigned q = q_first + 32 * b + r;
                    const bool q_live = q >= col_q0 && q < n0;
                    const unsigned long long live = __ballot(q_live);
#ifdef AKZ_MM_COLS_NOHIT  // measurement: the pass without the opposite direction's candidates
                    any = 0ull;
#endif
                    if ((any & live) != 0ull) {  // a row of the sub-tile may take one of the wave's queries as one of its two nearest
#ifdef AKZ_MM_COUNT
                        if (lane == 0) atomicAdd(&s_count[0], 1u);
#endif
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            // (wave-uniform tests, four accumulators at a time first: a scan of sixteen dependent scalar
                            //  compare-and-branch steps was ~300 cycles that the wave's workgroup then waited for at the barrier)
                            if (((hit[4 * g4] | hit[4 * g4 + 1] | hit[4 * g4 + 2] | hit[4 * g4 + 3]) & live) == 0ull) continue;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int i = 4 * g4 + k;
                                if ((hit[i] & live) == 0ull) continue;
                                if (acc[b][i] >= lim[i] && q_live) {
                                    const unsigned d = (unsigned)((kBits - (int)acc[b][i]) >> 1);
                                    if (d < threshold) {
                                        const size_t prow = (size_t)tile * MM_TR + 32u * sub + 4u * h + (unsigned)((i & 3) + 8 * (i >> 2));
#ifdef AKZ_MM_COLS_NOATOMIC  // measurement: candidates found, nothing entered
                                        if (d == 0xdeadbeefu) csecond[prow] = d;
#else
                                        // (the exchange's answer is looked at after the next commit, or when the lane's next
                                        //  candidate comes first: a wave that waited for it here held its workgroup at the barrier)
#ifdef AKZ_MM_COUNT
                                        atomicAdd(&s_count[1], 1u);
                                        if (d < csecond[prow]) atomicAdd(&s_count[3], 1u);
#endif
                                        if (__ballot(pend_row != 0xffffffffu) != 0ull) settle();
                                        const unsigned long long mine = ((unsigned long long)d << 32) | q;
                                        pend_old = atomicMin(cbest + prow, mine);
                                        pend_d = d;
                                        pend_row = (unsigned)prow;
#endif
                                    }
                                }
                            }
                        }
                    }
                }
                // The exact update.  best < limit <= second: the lane's 16 rows improve its second distance at least (see `limit`
                // in k_match_mfma).  Unless they also hold a new minimum that is all there is to do; a new minimum needs the
                // row of the first largest accumulator and the second largest one: keys 16 acc + (15 - i), exact in fp32,
                // through a top-2 network of v_max3 / v_med3 (~58 instructions, for about a third of the sub-tiles).  The
                // SIMD's vector issue is what the loop runs out of next to its matrix instructions (one MFMA holds it for 8
                // of its 32 cycles, every other instruction for 4): the former body -- integer keys and the padding tests of
                // a partial tile on every entry, ~110 instructions for 47..95 % of the sub-tiles -- cost 12 % of the kernel
                // at 89 816 x 89 816 and 22 % in the multi-set launch (round 4, profiles/r04_match_mutual.txt).
                if (bestf < (float)limit[b]) {
                    const unsigned tb = (unsigned)bestf, before = second[b];
                    if (tb < min_d[b]) {
                        float key[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) key[i] = __builtin_fmaf(acc[b][i], 16.0f, (float)(15 - i));
                        float M = __builtin_fmaxf(key[0], key[1]), S = __builtin_fminf(key[0], key[1]);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float ka = key[2 + 3 * g], kb = key[3 + 3 * g], kc = key[4 + 3 * g];
                            const float gm = __builtin_fmaxf(ka, __builtin_fmaxf(kb, kc)), gs = __builtin_amdgcn_fmed3f(ka, kb, kc);
                            S = __builtin_fmaxf(__builtin_fminf(M, gm), __builtin_fmaxf(S, gs));
                            M = __builtin_fmaxf(M, gm);
                        }
                        {
                            const float gm = __builtin_fmaxf(key[14], key[15]), gs = __builtin_fminf(key[14], key[15]);
                            S = __builtin_fmaxf(__builtin_fminf(M, gm), __builtin_fmaxf(S, gs));
                            M = __builtin_fmaxf(M, gm);
                        }
                        const int i1 = 15 - ((int)M & 15);
                        const unsigned ts = S < kPadKey ? 0xffffffffu : (unsigned)((kBits - ((int)S >> 4)) >> 1);
                        second[b] = min(min_d[b], ts);
                        min_d[b] = tb;
                        min_j[b] = j0 + (unsigned)((i1 & 3) + 8 * (i1 >> 2));
                    } else {
                        second[b] = tb;
                    }
                    if (second[b] < before) {
                        limit[b] = min(limit[b], second[b]);
                    }
                }
            }
        }
        if (AHEAD == 1 || step % AHEAD == AHEAD - 1 || tile + STEP >= t_end) __syncthreads();  // (uniform)
    }
    settle();
#ifdef AKZ_MM_COUNT
    __syncthreads();
    if (COLS && tid == 0)
        printf("COUNT %u %u %u %u %u\n", (t_end - t_begin) * 4u * 16u, s_count[0], s_count[1], s_count[2], s_count[3]);
#endif
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const unsigned o_min = __shfl_xor(min_d[b], 32, 64), o_sec = __shfl_xor(second[b], 32, 64), o_j = __shfl_xor(min_j[b], 32, 64);
        unsigned m = min_d[b], s2 = second[b], j = min_j[b];
        if (o_min < m || (o_min == m && o_min < threshold && o_j < j)) {
            s2 = min(o_sec, m);
            m = o_min;
            j = o_j;
        } else {
            s2 = min(s2, o_min);
        }
        const unsigned q = q_first + 32 * b + r;
        if (h == 0 && q < n0) {
            MatchRec rec;
            rec.min_d = m; rec.second_d = s2; rec.min_j = j; rec._pad = 0;
            out[(size_t)record * n0 + q] = rec;
        }
    }
}

// ---- the opposite direction's state and lists (COLS) ----------------------------------------------------------------
// seed records (the train rows as queries against the first col_q0 rows of the query image) -> cbest / csecond
__global__ void k_cols_init(const MatchRec* __restrict__ seed, unsigned rows, unsigned long long* __restrict__ cbest,
                            unsigned* __restrict__ csecond) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const MatchRec m = seed[i];
    cbest[i] = ((unsigned long long)m.min_d << 32) | m.min_j;
    csecond[i] = m.second_d;
}
// One workgroup per train set: its rows' (best, second) -> Lowe ratio^2 + threshold test (feature_matching.rs:61-63) and
// ordered compaction, as k_match_compact does for the query direction.  sets[k] = {first padded row, rows, first output}.
struct ColSet {
    unsigned row0, rows, out0;
};
__global__ void __launch_bounds__(1024) k_match_compact_cols(const unsigned long long* __restrict__ cbest,
                                                             const unsigned* __restrict__ csecond, const ColSet* __restrict__ sets,
                                                             unsigned threshold, double ratio2, akz_match* __restrict__ out,
                                                             unsigned long long* __restrict__ n_out) {
    constexpr int IT = 16;  // (rounds of 16 x 1024 rows, two barriers each: as k_match_compact)
    __shared__ unsigned s_cnt[IT][16];
    __shared__ unsigned s_base;
    const ColSet cs = sets[blockIdx.x];
    out += cs.out0;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (unsigned start = 0; start < cs.rows; start += IT * 1024u) {
        unsigned md[IT], mj[IT], keepmask = 0u;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const unsigned i = start + (unsigned)it * 1024u + threadIdx.x;
            bool keep = false;
            md[it] = mj[it] = 0u;
            if (i < cs.rows) {
                const unsigned long long b = cbest[(size_t)cs.row0 + i];
                const unsigned second = csecond[(size_t)cs.row0 + i];
                md[it] = (unsigned)(b >> 32);
                mj[it] = (unsigned)b;
                keep = ((double)md[it] < (double)second * ratio2) && (md[it] < threshold);
            }
            const unsigned long long bal = __ballot(keep);
            if (lane == 0) s_cnt[it][wave] = (unsigned)__popcll(bal);
            keepmask |= keep ? 1u << it : 0u;
        }
        __syncthreads();
        unsigned off = s_base;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            unsigned row = 0, below = 0;
#pragma unroll
            for (unsigned w = 0; w < 16; ++w) {
                const unsigned c = s_cnt[it][w];
                row += c;
                below += w < wave ? c : 0u;
            }
            const bool keep = (keepmask >> it) & 1u;
            const unsigned long long bal = __ballot(keep);
            if (keep) {
                akz_match o;
                o.index_0 = start + (unsigned)it * 1024u + threadIdx.x; o.index_1 = mj[it]; o.distance = (double)md[it];
                out[off + below + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = o;
            }
            off += row;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_base = off;
        __syncthreads();
    }
    if (threadIdx.x == 0) n_out[blockIdx.x] = s_base;
}

}  // namespace

namespace launch {

// rows of the unpacked image of a set of n descriptors (queries: whole workgroups; train: whole tiles)
uint32_t match_mfma_rows(uint32_t n, bool queries) {
    const uint32_t m = queries ? (uint32_t)std::max(MM_QB, MM4_QB) : (uint32_t)MM_TR;
    return (std::max<uint32_t>(n, 1) + m - 1) / m * m;
}
uint32_t match_mfma_chunks(uint32_t n0, uint32_t n1, uint32_t forced) {
    if (forced) {  // akz_debug_set_match_chunks
        const uint32_t tiles_e = (std::max<uint32_t>(n1, 1) + MM_TR - 1) / MM_TR;
        return std::max<uint32_t>(1, std::min<uint32_t>(forced, tiles_e));
    }
    const uint32_t qblocks = (std::max<uint32_t>(n0, 1) + MM_QB - 1) / MM_QB;
    const uint32_t tiles = (std::max<uint32_t>(n1, 1) + MM_TR - 1) / MM_TR;
    // One workgroup per CU is resident (16 waves); a launch runs in rounds of 256 workgroups, each of which first
    // loads its 512 queries (as much traffic as 512 train rows) and then walks its chunk.  Pick the chunk count
    // with the least estimated time: rounds x (tiles per chunk + 512 / rows per tile).
    constexpr uint32_t kQueryLoad = 512 / MM_TR;
    const uint32_t cmax = std::max<uint32_t>(1, std::min<uint32_t>(tiles / (kQueryLoad / 2), 128u));
    uint32_t best = 1;
    uint64_t best_cost = ~0ull;
    for (uint32_t c = 1; c <= cmax; ++c) {
        const uint64_t rounds = ((uint64_t)qblocks * c + 255) / 256;
        const uint64_t cost = rounds * ((tiles + c - 1) / c + kQueryLoad);
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}
void unpack_bits(hipStream_t s, const uint8_t* d, uint32_t n, uint32_t n_pad, bool query, uint8_t* out8, uint32_t* pop,
                 uint32_t* bound, uint32_t threshold, uint32_t n_bound, const uint32_t* d_tiles, bool fp4) {
    if (fp4) {
        hipLaunchKernelGGL(k_unpack_bits_fp4, dim3((n_pad + 15) / 16), dim3(256), 0, s, d, n, n_pad, out8, pop, bound, threshold, n_bound,
                           reinterpret_cast<const uint2*>(d_tiles));
        return;
    }
    hipLaunchKernelGGL(k_unpack_bits, dim3((n_pad + 3) / 4), dim3(256), 0, s, d, n, n_pad, query, out8, pop, bound, threshold,
                       n_bound, reinterpret_cast<const uint2*>(d_tiles), fp4);
}
void unpack_pair(hipStream_t s, const uint8_t* dq, uint32_t nq, uint32_t q_pad, uint8_t* outq, uint32_t* popq, uint32_t* bound, uint32_t threshold,
                 const uint8_t* dt, uint32_t nt, uint32_t t_pad, uint8_t* outt, uint32_t* popt, bool fp4) {
    if (fp4 && q_pad % 16 == 0) {
        hipLaunchKernelGGL(k_unpack_pair_fp4, dim3((q_pad + t_pad + 15) / 16), dim3(256), 0, s, dq, nq, q_pad, outq, popq, bound, threshold,
                           dt, nt, t_pad, outt, popt);
        return;
    }
    hipLaunchKernelGGL(k_unpack_pair, dim3((q_pad + t_pad + 3) / 4), dim3(256), 0, s, dq, nq, q_pad, outq, popq, bound, threshold, dt, nt,
                       t_pad, outt, popt, fp4);
}
uint32_t match_mfma_tile_rows() { return MM_TR; }
uint32_t match_mfma_query_block() { return MM_QB; }
// Chunks per train set of a multi-set launch: one workgroup per CU is resident, so a launch runs in rounds of 256
// workgroups; with one chunk per set 22 query blocks x 16 sets are 352 workgroups = 1.4 rounds that take as long as 2.
// Same estimate as match_mfma_chunks: rounds x (tiles per chunk + the query load).
uint32_t match_mfma_multi_chunks(uint32_t n0, uint32_t n_sets, uint32_t avg_tiles, uint32_t forced) {
    if (forced) return std::min(16u, forced);  // akz_debug_set_match_chunks
    const uint64_t qblocks = (std::max<uint32_t>(n0, 1) + MM_QB - 1) / MM_QB;
    constexpr uint32_t kQueryLoad = 512 / MM_TR;
    uint32_t best = 1;
    uint64_t best_cost = ~0ull;
    for (uint32_t c = 1; c <= 8 && (c == 1 || avg_tiles / c >= 2 * kQueryLoad); ++c) {
        const uint64_t rounds = (qblocks * n_sets * c + 255) / 256;
        const uint64_t cost = rounds * ((avg_tiles + c - 1) / c + kQueryLoad);
        if (cost < best_cost) {
            best_cost = cost;
            best = c;
        }
    }
    return best;
}
// One launch for a query set against SEVERAL train sets (padded image t8 as laid out by unpack_bits with a tile table):
// d_table holds n_chunks chunk descriptors (several per set, ascending rows); record r * n0 + q = top-2 of query q over
// the chunk with record index r.
void match_mfma_multi(hipStream_t s, const uint8_t* q8, const uint32_t* qpop, uint32_t n0, const uint8_t* t8,
                      const void* d_table, uint32_t n_chunks, uint32_t threshold, uint32_t* bound, MatchRec* d_out, bool fp4) {
    if (n0 == 0 || n_chunks == 0) return;
    if (fp4) {
        hipLaunchKernelGGL((k_match_fp4<AKZ_MM4_NB, AKZ_MM4_NT>), dim3((n0 + MM4_QB - 1) / MM4_QB, n_chunks), dim3(AKZ_MM4_NT), 0, s, q8, n0, t8, 0u, 0u,
                           threshold, bound, d_out, reinterpret_cast<const MatchChunk*>(d_table));
        return;
    }
    hipLaunchKernelGGL(k_match_mfma, dim3((n0 + MM_QB - 1) / MM_QB, n_chunks), dim3(MM_NT), 0, s, q8, qpop, n0, t8, 0u, 0u,
                       threshold, bound, d_out, reinterpret_cast<const MatchChunk*>(d_table));
}
// Both directions (FP4 form only).  match_cols_seed: the padded train image's rows as QUERIES against the first seed_rows
// (a multiple of the tile height, or all n0 if fewer) rows of the query image -> exact (best, second) of every train row
// over those queries in cbest / csecond; d_bound / d_seed: scratch of t_rows_pad u32 / MatchRec (t_rows_pad: the padded
// train rows rounded up to whole query blocks; both images are allocated to that).
#ifndef AKZ_MM_SEED_TILES
#define AKZ_MM_SEED_TILES 16
#endif
uint32_t match_cols_seed_rows(uint32_t n0) { return std::min<uint32_t>(n0, (uint32_t)AKZ_MM_SEED_TILES * MM_TR); }
void match_cols_seed(hipStream_t s, const uint8_t* q4, uint32_t n0, const uint8_t* t4, uint32_t t_rows, uint32_t threshold,
                     uint32_t* d_bound, MatchRec* d_seed, unsigned long long* cbest, uint32_t* csecond) {
    const uint32_t seed = match_cols_seed_rows(n0);
    if (t_rows == 0) return;
#if AKZ_MM4_BOUND
    (void)hipMemsetAsync(d_bound, 0xff, (size_t)match_mfma_rows(t_rows, true) * sizeof(uint32_t), s);  // no pruning bound yet
#endif
    const uint32_t tiles = (std::max<uint32_t>(seed, 1) + MM_TR - 1) / MM_TR;
    hipLaunchKernelGGL((k_match_fp4<AKZ_MM4_NB, AKZ_MM4_NT>), dim3((t_rows + MM4_QB - 1) / MM4_QB, 1), dim3(AKZ_MM4_NT), 0, s, t4, t_rows, q4,
                       seed, tiles, threshold, d_bound, d_seed, (const MatchChunk*)nullptr, (unsigned long long*)nullptr,
                       (unsigned*)nullptr, 0u);
    hipLaunchKernelGGL(k_cols_init, dim3((t_rows + 255) / 256), dim3(256), 0, s, d_seed, t_rows, cbest, csecond);
}
// the multi-set launch with the opposite direction riding along (queries from seed_rows on)
void match_fp4_multi_mutual(hipStream_t s, const uint8_t* q4, uint32_t n0, const uint8_t* t4, const void* d_table, uint32_t n_chunks,
                            uint32_t threshold, uint32_t* bound, MatchRec* d_out, unsigned long long* cbest, uint32_t* csecond) {
    if (n0 == 0 || n_chunks == 0) return;
    hipLaunchKernelGGL((k_match_fp4<AKZ_MM4_NB, AKZ_MM4_NT, true>), dim3((n0 + MM4_QB - 1) / MM4_QB, n_chunks), dim3(AKZ_MM4_NT), 0, s, q4, n0,
                       t4, 0u, 0u, threshold, bound, d_out, reinterpret_cast<const MatchChunk*>(d_table), cbest, csecond,
                       match_cols_seed_rows(n0));
}
void match_compact_cols(hipStream_t s, const unsigned long long* cbest, const uint32_t* csecond, const void* d_sets, uint32_t n_sets,
                        uint32_t threshold, double ratio2, akz_match* d_out, unsigned long long* d_n_out) {
    if (n_sets == 0) return;
    hipLaunchKernelGGL(k_match_compact_cols, dim3(n_sets), dim3(1024), 0, s, cbest, csecond, reinterpret_cast<const ColSet*>(d_sets),
                       threshold, ratio2, d_out, d_n_out);
}
// records of every query over `chunks` chunks of the train set: d_rec[chunk * n0 + query] (merged by match_compact)
void match_mfma(hipStream_t s, const uint8_t* q8, const uint32_t* qpop, uint32_t n0, const uint8_t* t8, uint32_t n1,
                uint32_t threshold, uint32_t* bound, uint32_t chunks, MatchRec* d_rec, bool fp4) {
    if (n0 == 0) return;
    const uint32_t tiles = (std::max<uint32_t>(n1, 1) + MM_TR - 1) / MM_TR;
    const uint32_t chunk_tiles = (tiles + chunks - 1) / chunks;
    if (fp4) {
        hipLaunchKernelGGL((k_match_fp4<AKZ_MM4_NB, AKZ_MM4_NT>), dim3((n0 + MM4_QB - 1) / MM4_QB, chunks), dim3(AKZ_MM4_NT), 0, s, q8, n0, t8, n1,
                           chunk_tiles, threshold, bound, d_rec, (const MatchChunk*)nullptr);
        return;
    }
    hipLaunchKernelGGL(k_match_mfma, dim3((n0 + MM_QB - 1) / MM_QB, chunks), dim3(MM_NT), 0, s, q8, qpop, n0, t8, n1,
                       chunk_tiles, threshold, bound, d_rec, (const MatchChunk*)nullptr);
}

}  // namespace launch
}  // namespace akz
