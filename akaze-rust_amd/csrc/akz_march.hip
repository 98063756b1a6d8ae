// The column-march kernels of the pyramid (gfx950): k_detector_march (below), k_level_march (level preparation + the
// level's first diffusion steps) and k_blur5_march (the level-0 blur).  They share the strip / band geometry, the
// range-checked buffer accesses ("global memory" section) and the one-LDS-exchange-per-row structure described here
// for the detector:
//
// Detector response of one level in ONE pass (detector_response.rs:8-55 + the extrema test of
// scale_space_extrema.rs:32-42, :80-87): Lsmooth -> Lx, Ly -> Lxx, Lyy, Lxy -> Ldet -> candidates, as a column march.
//
// A 256-thread workgroup owns a strip of 512 columns and marches down a band of rows; a thread owns TWO adjacent
// columns, so every arithmetic instruction is a packed f32 operation (v_pk_mul_f32 / v_pk_add_f32, IEEE per
// component, no FMA).  HBM sees each plane once: 4 B read and 12 (+12 with the second derivatives kept) written per
// pixel, plus the band / strip halos of the input (a few per cent); Lx and Ly are never read back.
//
//   * vertical taps (rows r-2S, r-S, r) come from REGISTER rings of the last 2S+1 rows — the row loop is unrolled by
//     the ring length, so every ring index is static (5 rings x (2S+1) rows x 2 columns: 90 VGPRs at S = 4);
//   * horizontal taps (columns x-S, x, x+S) come from LDS: each iteration every thread writes its two values of the
//     rows that the horizontal passes of this iteration read — Lsmooth row v, Lx / Ly row v-S-1, Ldet row v-2S-2 —
//     into a row buffer, ONE barrier, then reads its neighbours (buffers alternate, so no second barrier).  The
//     stages are skewed by one iteration each, which is what makes one barrier per row enough:
//
//        iteration t:  Lsmooth row v --H--> (Hm, Ho) ring --V--> Lx, Ly row u = v-S   (stored; into LDS next iteration)
//                      Lx, Ly row u-1 --H (LDS)--> (A, B, C) rings --V--> Lxx, Lyy, Lxy, Ldet row c = u-1-S (stored)
//                      Ldet rows c-2, c-1, c (registers) + row c-1's neighbours (LDS) --> extrema test of row c-1
//
// fill_border (types/image.rs:239-260) is reproduced exactly: a pass result at (x, y) is the interior result at
// (clamp(x,S,w-1-S), clamp(y,S,h-1-S)).  Columns: every thread reads its horizontal taps around its CLAMPED column, so
// every LDS position and ring slot holds the filled value of its own column.  Rows: the H pass of virtual row v reads
// input row clamp(v); the filled Lx / Ly rows above row S are row S (its stage-2 H result is entered into the S ring
// slots above it when it appears), the rows below h-1-S repeat row h-1-S (the stage-1 V result is held); border rows
// of the outputs are stored by the band that owns rows S / h-1-S.
//
// Arithmetic is the reference's: f32 mul then add, taps left to right starting from 0.0f; the off-axis Scharr taps
// [-1, 0.., 0, ..0, 1] are evaluated as (0.0f - a) + c (bit-identical for finite values).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "akz_internal.hpp"
#include "akz_pm_g2.hpp"

namespace akz {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));  // two pixels of a row, dword-aligned

constexpr int MT = 256;          // threads per workgroup (4 waves)
constexpr int MW = 2 * MT;       // columns of a strip, halo included
constexpr int HALO = 16;         // halo columns on each side (>= 2S+2)
constexpr int USE = MW - 2 * HALO;  // 480 columns owned per strip: strips start on 128-byte boundaries of a row (for the
                                    // usual widths), which is worth 30 % of HBM bandwidth against strips that do not
                                    // (tools/membw/marchbw: 3.8 -> 5.0 TB/s for this access shape, 6.0 with streaming stores)
constexpr int PAD = 8;           // floats left and right of an LDS row (>= S; even: position 0 stays 8-byte aligned)
constexpr int ROW = MW + 2 * PAD;
constexpr int CBUF = 32;         // extrema buffered per wave before one atomic reserves their list slots
#ifndef AKZ_MARCH_PF
#define AKZ_MARCH_PF 3
#endif

struct MarchGrid {
    int nstrips, nbands, band_rows;  // band_rows: interior rows per band
    int edge_rows;                   // level march: rows of the first band (the last one takes what is left); 0: band_rows
    int total;                       // workgroups with work; the grid is rounded up to a multiple of 8
};
// Workgroup i runs on XCD i mod 8.  Numbering the (image, band, strip) cells so that each XCD walks a CONTIGUOUS eighth
// of them keeps an image's strips -- which share their halo columns -- in one L2, and lets each XCD's address translation
// cover an eighth of the planes instead of all of them (measured: a launch on planes the previous launch did not touch
// is a third slower than a repeat on the same planes, and in the pyramid every launch is of the first kind).
#ifndef AKZ_MARCH_XCD
#define AKZ_MARCH_XCD 1
#endif
__device__ __forceinline__ int march_cell() {
#if AKZ_MARCH_XCD
    return (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3));
#else
    return (int)blockIdx.x;
#endif
}
// Interior rows [cs, ce) of a band -- the one place the kernels and the host-side check (launch::march_band_rows) get
// them from.  Detector / blur: uniform bands of band_rows rows from row S.  Level kernel: the first band has edge_rows
// rows, the last one takes what is left (bands at the image's first and last rows run the loop with the border cases,
// which is slower per row, and get fewer rows for it).
__host__ __device__ __forceinline__ void uniform_band_rows(const MarchGrid& g, int band, int S, int h, int* cs, int* ce) {
    *cs = S + band * g.band_rows;
    const int e = *cs + g.band_rows;
    *ce = e < h - S ? e : h - S;
}
__host__ __device__ __forceinline__ void level_band_rows(const MarchGrid& g, int band, int h, int* cs, int* ce) {
    *cs = 1 + (band == 0 ? 0 : g.edge_rows + (band - 1) * g.band_rows);
    const int e = *cs + (band == 0 ? g.edge_rows : g.band_rows);
    *ce = band == g.nbands - 1 ? h - 1 : (e < h - 1 ? e : h - 1);
}
struct MarchNms {
    unsigned level;
    float thr;
    int xlo, xhi, ylo, yhi;
    Candidate* cand;
    unsigned cap;
    unsigned* count;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ f2 tap_main(f2 a, f2 b, f2 c, float kn, float kwn) {
    const f2 z = {0.0f, 0.0f};
    return ((z + kn * a) + kwn * b) + kn * c;
}
__device__ __forceinline__ f2 tap_off(f2 a, f2 c) {
    const f2 z = {0.0f, 0.0f};
    return (z - a) + c;
}

__device__ __forceinline__ void cands_flush(Candidate* buf, unsigned& n, const MarchNms& nms, int lane) {
    if (n == 0) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(nms.count, n);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    const uint4* src = reinterpret_cast<const uint4*>(buf);
    uint4* dst = reinterpret_cast<uint4*>(nms.cand);
    for (unsigned e = (unsigned)lane; e < 2u * n; e += 64u)  // 16-byte halves of the 32-byte records
        if (base + (e >> 1) < nms.cap) dst[2 * (size_t)base + e] = src[e];
    __builtin_amdgcn_wave_barrier();
    n = 0;
}
// every lane offers the pixels of its pair whose bit is set in m; called by whole waves (wave-uniform control flow)
template <typename F>
__device__ __forceinline__ void cands_push(Candidate* buf, unsigned& n, unsigned m, const MarchNms& nms, int lane, F&& make) {
    while (__ballot(m != 0u)) {  // rare
        const bool have = m != 0u;
        const int i = have ? __ffs(m) - 1 : 0;
        m &= m - 1u;
        const unsigned long long b = __ballot(have);
        const unsigned nb = (unsigned)__popcll(b);
        if (n + nb > (unsigned)CBUF) cands_flush(buf, n, nms, lane);
        const unsigned before = (unsigned)__popcll(b & ((1ull << lane) - 1ull));
        if (nb <= (unsigned)CBUF) {
            if (have) buf[n + before] = make(i);
            n += nb;
        } else {  // more hits in one row of the wave than the buffer holds: two halves
            const bool lo = before < (unsigned)CBUF;
            if (have && lo) buf[before] = make(i);
            n = (unsigned)CBUF;
            __builtin_amdgcn_wave_barrier();
            cands_flush(buf, n, nms, lane);
            if (have && !lo) buf[before - CBUF] = make(i);
            n = nb - (unsigned)CBUF;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- global memory: raw buffer accesses --------------------------------------------------------------------------------
// Every access of the row loop is `buffer_{load,store}_dwordx2 data, voffset, rsrc, soffset`: rsrc = one image's plane
// (num_records = its bytes), voffset = the thread's column offset, soffset = the row's byte offset (scalar).  The
// hardware drops an access whose voffset is out of range (voffset >= num_records - soffset), so a thread that owns no
// column, and a row that is not stored this iteration, present VO_NONE instead of branching around the instruction.
// Two things follow.  (1) Stores and loads are straight-line code, no exec-mask branches and no per-access address
// arithmetic on the vector ALU.  (2) The compiler can COUNT the memory operations between a load and its use and waits
// with vmcnt(n) for exactly that load; with the stores under divergent branches it could not, and drained the whole
// queue (vmcnt(0)) once per row -- which bound these kernels by the memory latency of one row, not by bandwidth.
typedef unsigned u2 __attribute__((ext_vector_type(2)));
constexpr unsigned VO_NONE = 0x40000000u;  // out of range for every plane the kernels accept (<= 2^30 bytes per image)
constexpr int RSRC_FLAGS = 0x00020000;     // raw buffer, dword data format (gfx9 word 3)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const float* plane_of_image, int w, int h) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane_of_image), 0, w * h * 4, RSRC_FLAGS);
}

struct Cols {          // a thread's two columns
    int x0;            // image column of the first one (halo threads may lie outside the image)
    unsigned vo_st;    // store offset: byte offset of column x0 for the owner of the pair, VO_NONE otherwise
    unsigned vo_st1;   // odd widths: byte offset of column x0+1 if it is owned and inside the image, VO_NONE otherwise
    unsigned vo_ld;    // load offset: byte offset of column clamp(x0, 0, w-2)
    int ld_dup;        // 0: the loaded pair is (x0, x0+1); 1 / 2: both columns clamp to the pair's first / second column
    bool ld_edge;      // wave-uniform: some lane of the wave has ld_dup != 0
};
__device__ __forceinline__ Cols make_cols(int x0, bool inner, int w) {
    Cols C;
    C.x0 = x0;
    C.vo_st = (inner && x0 < w) ? (unsigned)x0 * 4u : VO_NONE;
    C.vo_st1 = (inner && x0 + 1 < w) ? (unsigned)(x0 + 1) * 4u : VO_NONE;
    const int xl = min(max(x0, 0), w - 2);
    C.vo_ld = (unsigned)xl * 4u;
    C.ld_dup = x0 < 0 ? 1 : (x0 > w - 2 ? 2 : 0);
    C.ld_edge = __ballot(C.ld_dup != 0) != 0ull;
    return C;
}
// A workgroup- or wave-uniform condition that is almost never true: kept a branch (the empty asm stops the compiler from
// turning the guarded register moves into selects that every row of every wave would execute).
__device__ __forceinline__ bool rare(bool c) {
    if (__builtin_expect(c, 0)) {
        asm volatile("");
        return true;
    }
    return false;
}
// the dwords at columns (xl, xl+1), xl = clamp(x0, 0, w-2), of row `ro` (byte offset of an existing row of the plane) ...
// (ODDW: rows of an odd-width image are not all 8-byte aligned, which 8-byte buffer accesses need: two dwords)
template <bool ODDW>
__device__ __forceinline__ f2 load_pair(__amdgpu_buffer_rsrc_t rs, unsigned ro, const Cols& C) {
    if (!ODDW) return __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)C.vo_ld, (int)ro, 0));
    f2 v;
    v.x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)C.vo_ld, (int)ro, 0));
    v.y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)C.vo_ld + 4, (int)ro, 0));
    return v;
}
// ... and what they mean for the thread: (in[clamp(x0)], in[clamp(x0+1)]).  Applied where the pair is CONSUMED, rows
// after the load was issued, so that nothing waits for the load early.
__device__ __forceinline__ f2 fix_pair(f2 v, const Cols& C) {
    if (rare(C.ld_edge)) {  // waves at the left / right edge of the image only
        if (C.ld_dup == 1) v.y = v.x;
        if (C.ld_dup == 2) v.x = v.y;
    }
    return v;
}
// Row r of N planes; `on` (workgroup-uniform): the row is stored at all.  A row that is not stored is presented as
// (row 0, VO_NONE): the row offset must stay inside the plane for the range check to mean anything.
// ODDW: a pair can straddle the right edge of the image, so the two columns are stored one by one.
template <int N, bool ODDW, bool NT = true>
__device__ __forceinline__ void store_rows(const __amdgpu_buffer_rsrc_t (&plane)[N], int r, int w, bool on, const Cols& C,
                                           const f2 (&v)[N]) {
    const unsigned ro = on ? (unsigned)r * ((unsigned)w * 4u) : 0u;
    // streaming stores (NT): the plane is written once and not read again soon; they also leave the Infinity Cache
    // alone, which ordinary stores fill
    constexpr int AUX = NT ? 2 : 0;
    if (!ODDW) {
        const unsigned vo = on ? C.vo_st : VO_NONE;
#pragma unroll
        for (int i = 0; i < N; ++i)
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v[i]), plane[i], (int)vo, (int)ro, AUX);
    } else {
        const unsigned vo0 = on ? C.vo_st : VO_NONE, vo1 = on ? C.vo_st1 : VO_NONE;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float a = v[i].x, b = v[i].y;  // (bit_cast of a vector ELEMENT picks element 0 with this compiler)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), plane[i], (int)vo0, (int)ro, AUX);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(b), plane[i], (int)vo1, (int)ro, AUX);
        }
    }
}
// interior row r of the output planes, plus (twice per image) the border rows that copy it
template <int S, int N, bool ODDW, bool NT = true>
__device__ __forceinline__ void store_filled(const __amdgpu_buffer_rsrc_t (&plane)[N], const Cols& C, int w, int h, int r, bool on,
                                             const f2 (&v)[N]) {
    store_rows<N, ODDW, NT>(plane, r, w, on, C, v);
    if (on && (r == S || r == h - 1 - S)) {  // workgroup-uniform, two rows of an image
        if (r == S) {
#pragma nounroll
            for (int y = 0; y < S; ++y) store_rows<N, ODDW, NT>(plane, y, w, true, C, v);
        }
        if (r == h - 1 - S) {
#pragma nounroll
            for (int y = h - S; y < h; ++y) store_rows<N, ODDW, NT>(plane, y, w, true, C, v);
        }
    }
}

template <int S, bool NMS, bool KEEP, bool ODDW>
__global__ void __launch_bounds__(MT, 3)
k_detector_march(const float* __restrict__ ls, float* __restrict__ lx_out, float* __restrict__ ly_out,
                 float* __restrict__ lxx_out, float* __restrict__ lyy_out, float* __restrict__ lxy_out,
                 float* __restrict__ ldet_out, int w, int h, MarchGrid g, float kn, float kwn, float quat, MarchNms nms) {
    constexpr int P = 2 * S + 1, NOUT = KEEP ? 4 : 1;
    static_assert(2 * S + 2 <= HALO, "strip halo");
    // input rows in flight ahead of the arithmetic; they sit in a ring of R slots, R a divisor of the unroll count P
    // (static indices), so no register is ever copied -- a copy would have to wait for the load it copies
    constexpr int R = P % 3 == 0 ? 3 : P;
    constexpr int PF = AKZ_MARCH_PF < R - 1 ? AKZ_MARCH_PF : R - 1;
    static_assert(S <= PAD, "LDS padding");
    __shared__ __attribute__((aligned(16))) float s_row[2][4][ROW];  // [buffer][Lsmooth, Lx, Ly, Ldet][PAD + position]
    __shared__ Candidate s_cands[NMS ? MT / 64 : 1][NMS ? CBUF : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    Candidate* const cbuf = s_cands[NMS ? wv : 0];
    unsigned cnum = 0;
    // (image, band, strip), strips fastest: workgroups that run side by side work on the same rows of the image
    // (integer division runs on the vector ALU: readfirstlane moves the uniform results back to scalar registers, so
    // that row addresses, loop bounds and branches derived from them stay scalar)
    const int per = g.nbands * g.nstrips;
    const int cell = march_cell();
    if (cell >= g.total) return;
    const int img = __builtin_amdgcn_readfirstlane(cell / per);
    const int rem = cell - img * per;
    const int band = __builtin_amdgcn_readfirstlane(rem / g.nstrips), strip = rem - band * g.nstrips;
    int cs, ce;  // interior rows of this band
    uniform_band_rows(g, band, S, h, &cs, &ce);
    if (cs >= ce) return;

    const int X0 = strip * USE - HALO;  // image column of position 0
    // threads 0..239 own positions 16..495 (so that every wave's stores start on a 512-byte boundary of the strip),
    // threads 240..247 the right halo, 248..255 the left halo
    const int p0 = (2 * tid + HALO) & (MW - 1);
    const bool inner = p0 >= HALO && p0 < MW - HALO;
    const Cols C = make_cols(X0 + p0, inner, w);
    // LDS read indices of tap -S of the two columns, evaluated at the clamped column (taps 0 and +S: + S, + 2S)
    const int i0 = PAD + clampi(clampi(C.x0, S, w - 1 - S) - X0, 0, MW - 1) - S;
    const int i1 = PAD + clampi(clampi(C.x0 + 1, S, w - 1 - S) - X0, 0, MW - 1) - S;
    const int wi = PAD + p0;
    unsigned xok = 0;  // bit i: column i of this thread may hold a candidate
    if (NMS) {
        if (inner && C.x0 < w && C.x0 >= nms.xlo && C.x0 <= nms.xhi) xok |= 1u;
        if (inner && C.x0 + 1 < w && C.x0 + 1 >= nms.xlo && C.x0 + 1 <= nms.xhi) xok |= 2u;
    }

    const size_t base = (size_t)img * (size_t)w * (size_t)h;
    const __amdgpu_buffer_rsrc_t in = plane_rsrc(ls + base, w, h);
    const __amdgpu_buffer_rsrc_t out1[2] = {plane_rsrc(lx_out + base, w, h), plane_rsrc(ly_out + base, w, h)};
    __amdgpu_buffer_rsrc_t out2[NOUT];
    out2[0] = plane_rsrc(ldet_out + base, w, h);
    if (KEEP) {
        out2[1] = plane_rsrc(lxx_out + base, w, h);
        out2[2] = plane_rsrc(lyy_out + base, w, h);
        out2[3] = plane_rsrc(lxy_out + base, w, h);
    }
    const __amdgpu_buffer_rsrc_t(&out2c)[NOUT] = out2;

    // the extrema test of rows [cs, ce) needs Ldet rows cs-1 .. ce, i.e. input rows cs-1-2S .. ce+2S
    const int v0 = cs - 1 - 2 * S;
    const int T = (ce - cs) + 4 * S + 3;
    auto feed = [&](int t) -> f2 {
        return load_pair<ODDW>(in, (unsigned)clampi(v0 + min(t, T - 1), S, h - 1 - S) * ((unsigned)w * 4u), C);
    };
    f2 q[R];
#pragma unroll
    for (int i = 0; i < PF; ++i) q[i] = feed(i);
    const f2 zero = {0.0f, 0.0f};
    f2 rHm[P], rHo[P], rA[P], rB[P], rC[P];
#pragma unroll
    for (int i = 0; i < P; ++i) rHm[i] = rHo[i] = rA[i] = rB[i] = rC[i] = zero;
    f2 lx_c = zero, ly_c = zero;  // Lx, Ly of row u-1 (computed by the previous iteration)
    f2 dm1 = zero, dm2 = zero;    // Ldet of rows c-1, c-2

    for (int t0 = 0; t0 < T; t0 += P) {
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int t = t0 + k;
            {  // (whole groups of P rows: the up to P-1 rows past T store nothing, their output rows lie past ce)
                const int v = v0 + t, u = v - S, u1 = u - 1, c = u1 - S;
                q[(k + PF) % R] = feed(t + PF);
                float* const buf = &s_row[t & 1][0][0];
                *reinterpret_cast<f2*>(buf + 0 * ROW + wi) = fix_pair(q[k % R], C);
                *reinterpret_cast<f2*>(buf + 1 * ROW + wi) = lx_c;
                *reinterpret_cast<f2*>(buf + 2 * ROW + wi) = ly_c;
                if (NMS) *reinterpret_cast<f2*>(buf + 3 * ROW + wi) = dm1;
                __syncthreads();
                const int k0 = (k + 1) % P, k1 = (k + P - S) % P;  // ring rows r-2S, r-S; slot k = row r
                // ---- stage 1: H pass of Lsmooth row v, V pass -> Lx, Ly of row u ----
                {
                    const float* r0 = buf + i0;
                    const float* r1 = buf + i1;
                    const f2 a = {r0[0], r1[0]}, b = {r0[S], r1[S]}, cc = {r0[2 * S], r1[2 * S]};
                    rHm[k] = tap_main(a, b, cc, kn, kwn);
                    rHo[k] = tap_off(a, cc);
                }
                f2 lx_n = lx_c, ly_n = ly_c;  // below row h-1-S the filled rows repeat it
                if (u <= h - 1 - S) {
                    lx_n = tap_off(rHm[k0], rHm[k]);
                    ly_n = tap_main(rHo[k0], rHo[k1], rHo[k], kn, kwn);
                }
                {
                    const f2 o1[2] = {lx_n, ly_n};
                    store_filled<S, 2, ODDW>(out1, C, w, h, u, u >= cs && u < ce, o1);
                }
                // ---- stage 2: H pass of Lx, Ly row u-1 (in LDS), V pass -> Lxx, Lyy, Lxy, Ldet of row c ----
                f2 A, B, Cc;
                {
                    const float* x0p = buf + ROW + i0;
                    const float* x1p = buf + ROW + i1;
                    const float* y0p = buf + 2 * ROW + i0;
                    const float* y1p = buf + 2 * ROW + i1;
                    const f2 xa = {x0p[0], x1p[0]}, xb = {x0p[S], x1p[S]}, xc = {x0p[2 * S], x1p[2 * S]};
                    const f2 ya = {y0p[0], y1p[0]}, yc = {y0p[2 * S], y1p[2 * S]};
                    A = tap_main(xa, xb, xc, kn, kwn);  // H_main(Lx)
                    B = tap_off(ya, yc);                // H_off(Ly)
                    Cc = tap_off(xa, xc);               // H_off(Lx)
                }
                rA[k] = A; rB[k] = B; rC[k] = Cc;
                if (u1 == S) {  // filled Lx / Ly rows 0..S-1 are row S
#pragma unroll
                    for (int d = 1; d <= S; ++d) {
                        rA[(k + P - d) % P] = A; rB[(k + P - d) % P] = B; rC[(k + P - d) % P] = Cc;
                    }
                }
                const f2 lxx = tap_off(rA[k0], rA[k]);
                const f2 lyy = tap_main(rB[k0], rB[k1], rB[k], kn, kwn);
                const f2 lxy = tap_main(rC[k0], rC[k1], rC[k], kn, kwn);
                const f2 det = ((lxx * lyy) - (lxy * lxy)) * quat;
                {
                    f2 o[NOUT];
                    o[0] = det;
                    if (KEEP) { o[1] = lxx; o[2] = lyy; o[3] = lxy; }
                    const f2(&oc)[NOUT] = o;
                    store_filled<S, NOUT, ODDW>(out2c, C, w, h, c, c >= cs && c < ce, oc);
                }
                // ---- extrema test of row c-1: dm1, between dm2 (above) and det (below) ----
                if (NMS) {
                    const int y = c - 1;
                    if (y >= cs && y < ce && y >= nms.ylo && y <= nms.yhi) {  // workgroup-uniform
                        const float* dr = buf + 3 * ROW + wi;
                        const float left = dr[-1], right = dr[2];
                        const bool h0 = (dm1.x > nms.thr) & (dm1.x > dm1.y) & (dm1.x > left) & (dm1.x > dm2.x) & (dm1.x > det.x);
                        const bool h1 = (dm1.y > nms.thr) & (dm1.y > right) & (dm1.y > dm1.x) & (dm1.y > dm2.y) & (dm1.y > det.y);
                        const unsigned m = ((h0 ? 1u : 0u) | (h1 ? 2u : 0u)) & xok;
                        cands_push(cbuf, cnum, m, nms, lane, [&](int i) {
                            Candidate cd;
                            cd.level = nms.level;
                            cd.idx = (unsigned)(y * w + C.x0 + i);
                            cd.v = i == 0 ? dm1.x : dm1.y;
                            cd.xp = i == 0 ? dm1.y : right;
                            cd.xm = i == 0 ? left : dm1.x;
                            cd.yp = i == 0 ? det.x : det.y;
                            cd.ym = i == 0 ? dm2.x : dm2.y;
                            cd.img = (unsigned)img;
                            return cd;
                        });
                    }
                    dm2 = dm1;
                    dm1 = det;
                }
                lx_c = lx_n;
                ly_c = ly_n;
            }
        }
    }
    if (NMS) cands_flush(cbuf, cnum, nms, lane);
}

// ---------------------------------------------------------------------------------------------------------------------
// One pyramid level that continues its octave, in ONE pass (lib.rs:92-118): the previous level's final Lt ->
// Lsmooth = gaussian_blur(Lt, 1.0) -> Scharr pair at scale 1 -> Lflow = pm_g2 -> the level's first N <= 4 FED steps
// (nonlinear_diffusion.rs:15-144) -> Lt (+ Lstep).  Same march as the detector: 512-column strips on 128-byte boundaries,
// two columns per thread in packed f32, vertical taps from register rings, horizontal neighbours from one LDS exchange
// per row.  HBM sees 4 B read and 12 (+4 with Lstep) written per pixel — the separate preparation and diffusion
// launches read 12 and write 12 (+4), and re-read their tile halos.
//
// Stages of iteration t (v = first virtual row + t), each one row behind the one it reads from:
//     Lt_prev row v    --H gauss-->  GH ring --V gauss-->  Lsmooth row v-1             (stored)
//     Lsmooth row v-2  --H scharr--> HM, HO rings --V scharr, pm_g2--> Lflow row v-3   (stored; into the c ring)
//     FED stage s = 1..N: L^s row v-4-s from L^(s-1) rows v-5-s .. v-3-s (stage 1: the Lt_prev ring), the c ring and
//     the x-pair sums c(x-1)+c(x), c(x)+c(x+1) of that row (formed once per row, delayed through a ring)
// Rings have 4 or 8 slots and the row loop is unrolled 8 times, so every ring index is static.
//
// Borders.  The two blurs: fill_border with half width 1, columns by evaluating at the clamped column, rows by entering
// row 1's H result into the slot of row 0 and by holding the V result below row h-2 (as in the detector).  FED has no
// filled border: missing-neighbour terms are dropped with the reference's expression per case (:84-137) — rows by
// workgroup-uniform branches, columns by selects in the few threads that touch column 0 or w-1.
// ---------------------------------------------------------------------------------------------------------------------
struct LevelTaus {
    float half_tau[4];  // 0.5f * (tau as f32) per fused step (nonlinear_diffusion.rs:67)
};
__device__ __forceinline__ f2 lds2(const float* p) { return f2{p[0], p[1]}; }  // two neighbouring floats, any alignment
__device__ __forceinline__ f2 tap3(f2 a, f2 b, f2 c, float k0, float k1, float k2) {
    const f2 z = {0.0f, 0.0f};
    return ((z + k0 * a) + k1 * b) + k2 * c;
}
__device__ __forceinline__ double octave_contrast(double k, unsigned pow) {  // lib.rs:84: one octave at a time, in f64
    for (unsigned i = 0; i < pow; ++i) k = k * 0.75;
    return k;
}

// HALF: the level opens an octave and `prev` is the previous octave's last Lt (pw x ph = 2w x 2h, pw a multiple of 4): its
// 2x2 mean (half_size, types/image.rs:102-118: (((0 + a(2x,2y)) + a(2x,2y+1)) + a(2x+1,2y)) + a(2x+1,2y+1)) / 4) is formed
// where a row is consumed, from two 16-byte loads per thread and row -- no separate k_half_size launch, no half-size
// plane written and read back.
typedef unsigned u4m __attribute__((ext_vector_type(4)));
template <int N, bool KEEPSTEP, bool ODDW, bool HALF = false>
__global__ void __launch_bounds__(MT, 3)
k_level_march(const float* __restrict__ prev, float* __restrict__ lsmooth_out, float* __restrict__ lflow_out,
              float* __restrict__ lt_out, float* __restrict__ lstep_out, int w, int h, MarchGrid g, float g0, float g1,
              float g2, float kn, float kwn, const double* __restrict__ d_k, unsigned k_pow, LevelTaus ht, int pw = 0, int ph = 0) {
    static_assert(!(HALF && ODDW), "the 2x2 mean is folded in for even level widths only");
    static_assert(N >= 1 && N <= 4, "fused diffusion steps");
    static_assert(N + 3 <= HALO, "strip halo");
    constexpr int NPL = 3 + N;  // LDS rows per iteration: Lt_prev, Lsmooth, Lflow, L^0 .. L^(N-1)
#ifndef AKZ_LEVEL_PF
#define AKZ_LEVEL_PF 2
#endif
    constexpr int PF = AKZ_LEVEL_PF;  // input rows in flight (<= 3: they sit in a ring of 4 slots with static indices)
    static_assert(PF >= 1 && PF <= 3, "prefetch ring");
    __shared__ __attribute__((aligned(16))) float s_row[2][NPL][ROW];
    const int tid = threadIdx.x;
    const int per = g.nbands * g.nstrips;
    const int cell = march_cell();
    if (cell >= g.total) return;
    const int img = __builtin_amdgcn_readfirstlane(cell / per);
    const int rem = cell - img * per;
    const int band = __builtin_amdgcn_readfirstlane(rem / g.nstrips), strip = rem - band * g.nstrips;
    // interior rows of Lsmooth / Lflow: the first band has edge_rows rows, the last one what is left (bands at the image's
    // first and last rows run the loop with the border cases, which is slower per row, and get fewer rows for it)
    int cs, ce;
    level_band_rows(g, band, h, &cs, &ce);
    if (cs >= ce) return;
    const int lt0 = cs == 1 ? 0 : cs, lt1 = ce == h - 1 ? h : ce;  // rows of Lt / Lstep (no filled border there)

    const int X0 = strip * USE - HALO;
    const int p0 = (2 * tid + HALO) & (MW - 1);
    const bool inner = p0 >= HALO && p0 < MW - HALO;
    const Cols C = make_cols(X0 + p0, inner, w);
    // LDS indices of tap -1 of the two columns of the blur / Scharr passes (evaluated at the clamped column)
    const int i0 = PAD + clampi(clampi(C.x0, 1, w - 2) - X0, 0, MW - 1) - 1;
    const int i1 = PAD + clampi(clampi(C.x0 + 1, 1, w - 2) - X0, 0, MW - 1) - 1;
    const int wi = PAD + p0;
    const bool edge_h = __ballot(i0 != wi - 1 || i1 != wi) != 0ull;  // wave-uniform: some lane reads at a clamped column
    // FED: which x-neighbours exist (only threads at column 0 / w-1 have one missing)
    const bool hxn0 = C.x0 > 0, hxp0 = C.x0 + 1 < w, hxn1 = C.x0 + 1 > 0, hxp1 = C.x0 + 2 < w;
    // wave-uniform, so that the selects below are a scalar branch that 99 % of the waves never take
    const bool edge_col = __ballot(!(hxn0 && hxp0 && hxn1 && hxp1)) != 0ull;

    const size_t base = (size_t)img * (size_t)w * (size_t)h;
    const __amdgpu_buffer_rsrc_t in = HALF ? plane_rsrc(prev + (size_t)img * (size_t)pw * (size_t)ph, pw, ph) : plane_rsrc(prev + base, w, h);
    const __amdgpu_buffer_rsrc_t o_ls[1] = {plane_rsrc(lsmooth_out + base, w, h)};
    const __amdgpu_buffer_rsrc_t o_lf[1] = {plane_rsrc(lflow_out + base, w, h)};
    __amdgpu_buffer_rsrc_t o_lt[KEEPSTEP ? 2 : 1];
    o_lt[0] = plane_rsrc(lt_out + base, w, h);
    if (KEEPSTEP) o_lt[1] = plane_rsrc(lstep_out + base, w, h);
    const __amdgpu_buffer_rsrc_t(&o_ltc)[KEEPSTEP ? 2 : 1] = o_lt;
    const double kc = octave_contrast(d_k[img], k_pow);
    const double inverse_k = 1.0 / (kc * kc);

    // Lt row r needs Lt_prev rows r-N-2 .. r+N+2 (N steps, the c ring one row wider, c itself two blurs of half width 1)
    const int v0 = lt0 - N - 2;
    const int T = (lt1 - lt0) + 2 * N + 6;
    struct Fed {  // a row as loaded: the pair itself, or (HALF) the 2 x 4 source pixels of the pair
        f2 v;
        u4m top, bot;
    };
    auto feed = [&](int t) -> Fed {
        Fed f;
        const unsigned row = (unsigned)clampi(v0 + min(t, T - 1), 0, h - 1);
        if constexpr (HALF) {
            const unsigned ro = 2u * row * ((unsigned)pw * 4u);
            f.top = __builtin_amdgcn_raw_buffer_load_b128(in, (int)(2u * C.vo_ld), (int)ro, 0);
            f.bot = __builtin_amdgcn_raw_buffer_load_b128(in, (int)(2u * C.vo_ld), (int)(ro + (unsigned)pw * 4u), 0);
        } else {
            f.v = load_pair<ODDW>(in, row * ((unsigned)w * 4u), C);
        }
        return f;
    };
    auto pair_of = [&](const Fed& f) -> f2 {
        if constexpr (HALF) {
            float a = 0.0f, b = 0.0f;
            a = a + __uint_as_float(f.top.x); a = a + __uint_as_float(f.bot.x); a = a + __uint_as_float(f.top.y); a = a + __uint_as_float(f.bot.y);
            b = b + __uint_as_float(f.top.z); b = b + __uint_as_float(f.bot.z); b = b + __uint_as_float(f.top.w); b = b + __uint_as_float(f.bot.w);
            return f2{a / 4.0f, b / 4.0f};
        } else {
            return f.v;
        }
    };
    Fed q[4];
#pragma unroll
    for (int i = 0; i < PF; ++i) q[i] = feed(i);
    const f2 zero = {0.0f, 0.0f};
    f2 LP[8], CR[8];              // Lt_prev rows v .. v-6; Lflow (filled) rows v-3 .. v-5-N
    f2 GH[4], HM[4], HO[4];       // H results of the last three rows of the two blurs
    f2 SXW[4], SXE[4];            // c(x-1)+c(x), c(x)+c(x+1) of rows v-5 .. v-4-N
    f2 W[N > 1 ? N - 1 : 1][4];   // W[s-2]: L^(s-1) rows of stage s >= 2
#pragma unroll
    for (int i = 0; i < 8; ++i) LP[i] = CR[i] = zero;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        GH[i] = HM[i] = HO[i] = SXW[i] = SXE[i] = zero;
#pragma unroll
        for (int j = 0; j < (N > 1 ? N - 1 : 1); ++j) W[j][i] = zero;
    }
    f2 ls_c = zero, lf_c = zero, gh_c = zero;  // Lsmooth row v-2, Lflow row v-4, H gauss of row v-1 (previous iteration's results)

    // The row loop runs whole groups of 8 (static ring indices); the up to 7 rows past T store nothing: their Lsmooth,
    // Lflow and Lt rows lie past ce / lt1.
    auto rows = [&](auto mid_tag) {
    constexpr bool MID = decltype(mid_tag)::value;
    for (int t0 = 0; t0 < T; t0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int t = t0 + k;
            {
                const int v = v0 + t;
                q[(k + PF) & 3] = feed(t + PF);
                float* const buf = &s_row[t & 1][0][0];
                LP[k & 7] = fix_pair(pair_of(q[k & 3]), C);
                *reinterpret_cast<f2*>(buf + 0 * ROW + wi) = LP[k & 7];
                *reinterpret_cast<f2*>(buf + 1 * ROW + wi) = ls_c;
                *reinterpret_cast<f2*>(buf + 2 * ROW + wi) = CR[(k - 5) & 7];
                *reinterpret_cast<f2*>(buf + 3 * ROW + wi) = LP[(k - 5) & 7];
#pragma unroll
                for (int s = 2; s <= N; ++s) *reinterpret_cast<f2*>(buf + (2 + s) * ROW + wi) = W[s - 2][(k - 1) & 3];
                __syncthreads();
                // ---- gaussian_blur(Lt, 1.0): H pass of row v, V pass -> Lsmooth row u = v-1 ----
                f2 gh = gh_c;  // below row h-2 the filled rows repeat it
                if (MID || v <= h - 2) {
                    f2 a, b, c;
                    if (!rare(edge_h)) {  // the neighbours as two (odd-aligned) pairs, the centre from registers
                        a = lds2(buf + wi - 1); b = LP[k & 7]; c = lds2(buf + wi + 1);
                    } else {
                        const float* r0 = buf + i0;
                        const float* r1 = buf + i1;
                        a = f2{r0[0], r1[0]}; b = f2{r0[1], r1[1]}; c = f2{r0[2], r1[2]};
                    }
                    gh = tap3(a, b, c, g0, g1, g2);
                }
                gh_c = gh;
                GH[k & 3] = gh;
                if (!MID && rare(v == 1)) GH[(k - 1) & 3] = gh;  // filled row 0 is row 1
                const int u = v - 1;
                f2 ls_n = ls_c;
                if (MID || u <= h - 2) ls_n = tap3(GH[(k - 2) & 3], GH[(k - 1) & 3], GH[k & 3], g0, g1, g2);
                {
                    const f2 o[1] = {ls_n};
                    if (MID) store_rows<1, ODDW>(o_ls, u, w, u >= cs && u < ce, C, o);
                    else store_filled<1, 1, ODDW>(o_ls, C, w, h, u, u >= cs && u < ce, o);
                }
                // ---- Scharr pair at scale 1 of Lsmooth row u-1 (in LDS), V pass -> Lx1, Ly1 -> Lflow row c = v-3 ----
                {
                    f2 a, b, c;
                    if (!rare(edge_h)) {
                        a = lds2(buf + ROW + wi - 1); b = ls_c; c = lds2(buf + ROW + wi + 1);
                    } else {
                        const float* r0 = buf + ROW + i0;
                        const float* r1 = buf + ROW + i1;
                        a = f2{r0[0], r1[0]}; b = f2{r0[1], r1[1]}; c = f2{r0[2], r1[2]};
                    }
                    HM[k & 3] = tap_main(a, b, c, kn, kwn);
                    HO[k & 3] = tap_off(a, c);
                }
                if (!MID && rare(u - 1 == 1)) {  // filled row 0 is row 1
                    HM[(k - 1) & 3] = HM[k & 3];
                    HO[(k - 1) & 3] = HO[k & 3];
                }
                const int c = v - 3;
                f2 lf_n = lf_c;
                if (MID || c <= h - 2) {
                    const f2 lx1 = tap_off(HM[(k - 2) & 3], HM[k & 3]);
                    const f2 ly1 = tap_main(HO[(k - 2) & 3], HO[(k - 1) & 3], HO[k & 3], kn, kwn);
                    lf_n = f2{pm_g2_px(lx1.x, ly1.x, inverse_k), pm_g2_px(lx1.y, ly1.y, inverse_k)};
                }
                CR[(k - 3) & 7] = lf_n;
                if (!MID && rare(c == 1)) CR[(k - 4) & 7] = lf_n;  // filled row 0 is row 1
                {
                    const f2 o[1] = {lf_n};
                    if (MID) store_rows<1, ODDW>(o_lf, c, w, c >= cs && c < ce, C, o);
                    else store_filled<1, 1, ODDW>(o_lf, C, w, h, c, c >= cs && c < ce, o);
                }
                // ---- x-pair sums of the Lflow row the first FED stage works on (row v-5, in LDS) ----
                {
                    const float* cr = buf + 2 * ROW + wi;
                    const f2 cc = CR[(k - 5) & 7];
                    SXW[k & 3] = lds2(cr - 1) + cc;  // {c(x0-1) + c(x0), c(x0) + c(x0+1)}
                    SXE[k & 3] = cc + lds2(cr + 1);  // {c(x0) + c(x0+1), c(x0+1) + c(x0+2)}
                }
                // ---- FED stages: L^s row r = v-4-s ----
                f2 lnew = zero, st = zero;
#pragma unroll
                for (int s = 1; s <= N; ++s) {
                    const int r = v - 4 - s;
                    f2 Lm, Lc, Lp;
                    if (s == 1) {
                        Lm = LP[(k - 6) & 7]; Lc = LP[(k - 5) & 7]; Lp = LP[(k - 4) & 7];
                    } else {
                        W[s - 2][k & 3] = lnew;  // L^(s-1) row r+1, produced by stage s-1 just now
                        Lm = W[s - 2][(k - 2) & 3]; Lc = W[s - 2][(k - 1) & 3]; Lp = lnew;
                    }
                    const float* lr = buf + (2 + s) * ROW + wi;  // L^(s-1) row r with its neighbours
                    const f2 Lw = lds2(lr - 1), Le = lds2(lr + 1);  // {L(x0-1), L(x0)}, {L(x0+1), L(x0+2)}
                    const f2 cN = CR[(k - 5 - s) & 7], cC = CR[(k - 4 - s) & 7], cS = CR[(k - 3 - s) & 7];
                    const f2 SU = cN + cC, SV = cC + cS;
                    const f2 XFW = SXW[(k - (s - 1)) & 3] * (Lc - Lw);  // x_neg: (c_W + c) * (L - L_W)
                    const f2 XFE = SXE[(k - (s - 1)) & 3] * (Le - Lc);  // x_pos: (c + c_E) * (L_E - L)
                    f2 tx;
                    if (!edge_col) {
                        tx = XFE - XFW;
                    } else {  // columns 0 / w-1: the missing term is dropped (:94-102, :122-137)
                        tx.x = hxp0 ? (hxn0 ? XFE.x - XFW.x : XFE.x) : -XFW.x;
                        tx.y = hxp1 ? (hxn1 ? XFE.y - XFW.y : XFE.y) : -XFW.y;
                    }
                    f2 tt;
                    if (MID || r + 1 < h) {                // workgroup-uniform
                        tt = tx + SV * (Lp - Lc);          // + y_pos
                        if (MID || r > 0) tt = tt - SU * (Lc - Lm);  // - y_neg
                    } else {
                        tt = tx + SU * (Lm - Lc);          // last row: y_pos taken towards y-1 (:104-119)
                    }
                    st = ht.half_tau[s - 1] * tt;
                    lnew = Lc + st;
                }
                {
                    const int r = v - 4 - N;
                    {
                        f2 o[KEEPSTEP ? 2 : 1];
                        o[0] = lnew;
                        if constexpr (KEEPSTEP) o[1] = st;
                        const f2(&oc)[KEEPSTEP ? 2 : 1] = o;
                        store_rows<KEEPSTEP ? 2 : 1, ODDW>(o_ltc, r, w, r >= lt0 && r < lt1, C, oc);
                    }
                }
                ls_c = ls_n;
                lf_c = lf_n;
            }
        }
    }
    };
    // Bands that stay clear of the image's first and last rows (every row they touch, the up to 7 surplus rows included)
    // run a copy of the loop without the border cases: no held or back-filled rows, every diffusion row has both y
    // neighbours.
    const int v_last = v0 + ((T + 7) & ~7) - 1;
    if (v0 >= N + 5 && v_last <= h - 2) rows(std::true_type{});
    else rows(std::false_type{});
}

// ---------------------------------------------------------------------------------------------------------------------
// The two passes of compute_contrast_factor (contrast_factor.rs:18-71) as marches: the front of k_level_march --
// gaussian_blur(in, 1.0), the scale-1 Scharr pair -- with nothing stored; over the interior pixels (rows 1..h-2,
// columns 1..w-2, :33-35) MODE 1 takes the maximum of lx^2 + ly^2 in f64 (sqrt is monotone and correctly rounded:
// max sqrt(s) == sqrt(max s)), MODE 2 the histogram of floor(nbins * (sqrt(s) / hmax)) (:49-57).  The bin of a pixel
// is a non-decreasing function of s: a guess from the hardware's approximate square root, off by one bin at most, is
// settled by the exact thresholds of the two neighbouring bins (k_contrast_thresholds, akz_stream.hip) instead of a
// correctly rounded f64 square root and division per pixel.  Two columns per thread, packed arithmetic, one LDS
// exchange per row.  Both passes stay bound by instruction issue (about 50 instructions per pixel, a third of them
// f64): 82 + 125 us per 32 x 1080p against 111 + 135 for k_prep_stream<false, 1 / 2>.
// ---------------------------------------------------------------------------------------------------------------------
struct ContrastMarchArgs {
    unsigned long long* hmax_bits;  // per image, non-negative f64 as its bit pattern (orders like the value)
    unsigned* hist;                 // per image, nbins counters
    unsigned nbins;
    const double* thr;              // MODE 2: per image, nbins + 1 bin thresholds on lx^2 + ly^2
};
constexpr int CM_COPIES = 2;  // sub-histograms per wave (lanes spread over them) against same-bin conflicts
template <int MODE, bool ODDW>
__global__ void __launch_bounds__(MT, 8)
k_contrast_march(const float* __restrict__ in_plane, int w, int h, MarchGrid g, float g0, float g1, float g2, float kn,
                 float kwn, ContrastMarchArgs ca) {
    constexpr int PF = 2;
    __shared__ __attribute__((aligned(16))) float s_row[2][2][ROW];  // per iteration: the input row, Lsmooth row v-2
    extern __shared__ __attribute__((aligned(16))) unsigned s_chist[];  // MODE 2: [wave][copy][bin], then [wave][thresholds]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = g.nbands * g.nstrips;
    const int cell = march_cell();
    if (cell >= g.total) return;
    const int img = __builtin_amdgcn_readfirstlane(cell / per);
    const int rem = cell - img * per;
    const int band = __builtin_amdgcn_readfirstlane(rem / g.nstrips), strip = rem - band * g.nstrips;
    int cs, ce;
    level_band_rows(g, band, h, &cs, &ce);
    if (cs >= ce) return;
    const int lt0 = cs == 1 ? 0 : cs, lt1 = ce == h - 1 ? h : ce;

    const int X0 = strip * USE - HALO;
    const int p0 = (2 * tid + HALO) & (MW - 1);
    const bool inner = p0 >= HALO && p0 < MW - HALO;
    const Cols C = make_cols(X0 + p0, inner, w);
    const int i0 = PAD + clampi(clampi(C.x0, 1, w - 2) - X0, 0, MW - 1) - 1;
    const int i1 = PAD + clampi(clampi(C.x0 + 1, 1, w - 2) - X0, 0, MW - 1) - 1;
    const int wi = PAD + p0;
    const bool edge_h = __ballot(i0 != wi - 1 || i1 != wi) != 0ull;
    // the thread's columns that are interior pixels it owns
    const bool own0 = inner && C.x0 >= 1 && C.x0 <= w - 2, own1 = inner && C.x0 + 1 >= 1 && C.x0 + 1 <= w - 2;

    const __amdgpu_buffer_rsrc_t in = plane_rsrc(in_plane + (size_t)img * (size_t)w * (size_t)h, w, h);
    double gmax = 0.0, bin_scale = 0.0;
    unsigned* myhist = nullptr;
    const double* mythr = nullptr;
    if (MODE == 2) {
        const double hmax = __longlong_as_double((long long)ca.hmax_bits[img]);
        unsigned* wh = s_chist + (size_t)wv * CM_COPIES * ca.nbins;
        for (unsigned b = (unsigned)lane; b < CM_COPIES * ca.nbins; b += 64u) wh[b] = 0u;  // wave-private: no barrier needed
        myhist = wh + (lane & (CM_COPIES - 1)) * ca.nbins;
        // the image's bin thresholds, one copy per workgroup (behind the histograms: an even number of words, so 8-byte
        // aligned for every bin count); the first barrier of the row loop orders these writes before the first read
        double* wt = reinterpret_cast<double*>(s_chist + (size_t)(MT / 64) * CM_COPIES * ca.nbins);
        for (unsigned b = (unsigned)tid; b <= ca.nbins; b += MT) wt[b] = ca.thr[(size_t)img * (ca.nbins + 1) + b];
        mythr = wt;
        bin_scale = (double)ca.nbins * (1.0 / hmax);
    }

    const int v0 = lt0 - 2;
    const int T = (lt1 - lt0) + 6;
    auto feed = [&](int t) -> f2 {
        return load_pair<ODDW>(in, (unsigned)clampi(v0 + min(t, T - 1), 0, h - 1) * ((unsigned)w * 4u), C);
    };
    f2 q[4];
#pragma unroll
    for (int i = 0; i < PF; ++i) q[i] = feed(i);
    const f2 zero = {0.0f, 0.0f};
    f2 GH[4], HM[4], HO[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) GH[i] = HM[i] = HO[i] = zero;
    f2 ls_c = zero, gh_c = zero;  // Lsmooth row v-2, H gauss of row v-1 (the previous iteration's results)

    auto take = [&](double ss) {
        if (MODE == 1) {
            if (ss > gmax) gmax = ss;
        } else if (ss != 0.0) {  // sqrt(ss) != 0.0 (contrast_factor.rs:51)
            const double ga = __builtin_amdgcn_sqrt(ss) * bin_scale;
            const unsigned gb = min(ga > 0.0 ? (unsigned)ga : 0u, ca.nbins - 1u);  // NaN -> 0
            unsigned b = gb;
            if (ss < mythr[gb]) b = gb - 1u;
            else if (ss >= mythr[gb + 1u]) b = gb + 1u;
            atomicAdd(&myhist[b], 1u);
        }
    };
    auto rows = [&](auto mid_tag) {
    constexpr bool MID = decltype(mid_tag)::value;
    for (int t0 = 0; t0 < T; t0 += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = t0 + k;
            const int v = v0 + t;
            q[(k + PF) & 3] = feed(t + PF);
            float* const buf = &s_row[t & 1][0][0];
            const f2 lp = fix_pair(q[k & 3], C);
            *reinterpret_cast<f2*>(buf + 0 * ROW + wi) = lp;
            *reinterpret_cast<f2*>(buf + 1 * ROW + wi) = ls_c;
            __syncthreads();
            // ---- gaussian_blur(in, 1.0): H pass of row v, V pass -> Lsmooth row u = v-1 ----
            f2 gh = gh_c;  // below row h-2 the filled rows repeat it
            if (MID || v <= h - 2) {
                f2 a, b, c;
                if (!rare(edge_h)) {
                    a = lds2(buf + wi - 1); b = lp; c = lds2(buf + wi + 1);
                } else {
                    const float* r0 = buf + i0;
                    const float* r1 = buf + i1;
                    a = f2{r0[0], r1[0]}; b = f2{r0[1], r1[1]}; c = f2{r0[2], r1[2]};
                }
                gh = tap3(a, b, c, g0, g1, g2);
            }
            gh_c = gh;
            GH[k & 3] = gh;
            if (!MID && rare(v == 1)) GH[(k - 1) & 3] = gh;  // filled row 0 is row 1
            const int u = v - 1;
            f2 ls_n = ls_c;
            if (MID || u <= h - 2) ls_n = tap3(GH[(k - 2) & 3], GH[(k - 1) & 3], GH[k & 3], g0, g1, g2);
            // ---- Scharr pair at scale 1 of Lsmooth row u-1 (in LDS), V pass -> lx, ly of row c = v-3 ----
            {
                f2 a, b, c;
                if (!rare(edge_h)) {
                    a = lds2(buf + ROW + wi - 1); b = ls_c; c = lds2(buf + ROW + wi + 1);
                } else {
                    const float* r0 = buf + ROW + i0;
                    const float* r1 = buf + ROW + i1;
                    a = f2{r0[0], r1[0]}; b = f2{r0[1], r1[1]}; c = f2{r0[2], r1[2]};
                }
                HM[k & 3] = tap_main(a, b, c, kn, kwn);
                HO[k & 3] = tap_off(a, c);
            }
            if (!MID && rare(u - 1 == 1)) {  // filled row 0 is row 1
                HM[(k - 1) & 3] = HM[k & 3];
                HO[(k - 1) & 3] = HO[k & 3];
            }
            const int c = v - 3;
            if (c >= cs && c < ce) {  // workgroup-uniform: the band's interior rows
                const f2 lx1 = tap_off(HM[(k - 2) & 3], HM[k & 3]);
                const f2 ly1 = tap_main(HO[(k - 2) & 3], HO[(k - 1) & 3], HO[k & 3], kn, kwn);
                if (own0) {
                    const double dx = (double)lx1.x, dy = (double)ly1.x;
                    take(dx * dx + dy * dy);
                }
                if (own1) {
                    const double dx = (double)lx1.y, dy = (double)ly1.y;
                    take(dx * dx + dy * dy);
                }
            }
            ls_c = ls_n;
        }
    }
    };
    const int v_last = v0 + ((T + 3) & ~3) - 1;
    if (v0 >= 5 && v_last <= h - 2) rows(std::true_type{});
    else rows(std::false_type{});

    if (MODE == 1) {
        unsigned long long bits = (unsigned long long)__double_as_longlong(sqrt(gmax));
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(bits, o, 64);
            bits = other > bits ? other : bits;
        }
        if (lane == 0 && bits != 0ull) atomicMax(ca.hmax_bits + img, bits);
    } else {
        const unsigned* wh = s_chist + (size_t)wv * CM_COPIES * ca.nbins;
        for (unsigned b = (unsigned)lane; b < ca.nbins; b += 64u) {
            unsigned v = 0;
            for (int cpy = 0; cpy < CM_COPIES; ++cpy) v += wh[cpy * ca.nbins + b];
            if (v) atomicAdd(&ca.hist[(size_t)img * ca.nbins + b], v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gaussian_blur with a dense 5-tap kernel as a march (types/image.rs:374-380: V(H(in)), fill_border after each pass):
// the level-0 blur of the pyramid (sigma 1.6) for large batches.  T = uint8_t folds in create_unit_float_image
// (types/image.rs:136): the 256 possible values of `f32::from(v) * 1f32 / 255f32` are tabulated once per workgroup
// with that very expression.  Per iteration: the input row v (two pixels per thread, converted where they are consumed)
// goes to LDS, one barrier, H pass from two even-aligned and two odd-aligned LDS pairs next to the thread's own pair,
// V pass over a 5-slot ring -> row v-2 stored.  The plane is stored with ordinary stores: the contrast passes read it
// next and find most of it in the Infinity Cache.
// ---------------------------------------------------------------------------------------------------------------------
struct BlurTaps5 {
    float k[5];
};
__device__ __forceinline__ f2 tap5(f2 a, f2 b, f2 c, f2 d, f2 e, const BlurTaps5& t) {
    const f2 z = {0.0f, 0.0f};
    return ((((z + t.k[0] * a) + t.k[1] * b) + t.k[2] * c) + t.k[3] * d) + t.k[4] * e;
}
template <typename T, bool ODDW>
__global__ void __launch_bounds__(MT, 8)
k_blur5_march(const T* __restrict__ in, float* __restrict__ out, int w, int h, MarchGrid g, BlurTaps5 tp) {
    constexpr int S = 2, P = 5, R = 5, PF = 3;
    constexpr bool U8 = sizeof(T) == 1;
    __shared__ __attribute__((aligned(16))) float s_row[2][ROW];
    __shared__ float s_lut[256];
    const int tid = threadIdx.x;
    if (U8) s_lut[tid & 255] = ((float)(tid & 255) * 1.0f) / 255.0f;  // MT == 256; made visible by the loop's first barrier
    const int per = g.nbands * g.nstrips;
    const int cell = march_cell();
    if (cell >= g.total) return;
    const int img = __builtin_amdgcn_readfirstlane(cell / per);
    const int rem = cell - img * per;
    const int band = __builtin_amdgcn_readfirstlane(rem / g.nstrips), strip = rem - band * g.nstrips;
    int cs, ce;  // interior rows of this band
    uniform_band_rows(g, band, S, h, &cs, &ce);
    if (cs >= ce) return;

    const int X0 = strip * USE - HALO;
    const int p0 = (2 * tid + HALO) & (MW - 1);
    const bool inner = p0 >= HALO && p0 < MW - HALO;
    const Cols C = make_cols(X0 + p0, inner, w);
    // LDS read indices of tap -2 of the two columns, evaluated at the clamped column
    const int i0 = PAD + clampi(clampi(C.x0, S, w - 1 - S) - X0, 0, MW - 1) - S;
    const int i1 = PAD + clampi(clampi(C.x0 + 1, S, w - 1 - S) - X0, 0, MW - 1) - S;
    const int wi = PAD + p0;
    const bool edge_h = __ballot(i0 != wi - S || i1 != wi + 1 - S) != 0ull;  // some lane reads at a clamped column

    const size_t base = (size_t)img * (size_t)w * (size_t)h;
    // u8 input: the same range-checked addressing in bytes
    const __amdgpu_buffer_rsrc_t rin = U8 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(in + base), 0, w * h, RSRC_FLAGS)
                                          : plane_rsrc(reinterpret_cast<const float*>(in + base), w, h);
    const __amdgpu_buffer_rsrc_t rout[1] = {plane_rsrc(out + base, w, h)};
    const int v0 = cs - S, T_ = (ce - cs) + 2 * S;  // H rows v0 .. v0 + T_ - 1
    typedef typename std::conditional<U8, unsigned, f2>::type Raw;
    auto feed = [&](int t) -> Raw {
        const int row = clampi(v0 + min(t, T_ - 1), S, h - 1 - S);  // fill_border after the H pass: its rows 0..S-1 are row S
        if constexpr (U8) {
            const unsigned vo = C.vo_ld >> 2;  // byte offset of column clamp(x0, 0, w-2)
            if (!ODDW) return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rin, (int)vo, row * w, 0);
            const unsigned a = __builtin_amdgcn_raw_buffer_load_b8(rin, (int)vo, row * w, 0);
            const unsigned b = __builtin_amdgcn_raw_buffer_load_b8(rin, (int)vo + 1, row * w, 0);
            return (a & 255u) | ((b & 255u) << 8);
        } else {
            return load_pair<ODDW>(rin, (unsigned)row * ((unsigned)w * 4u), C);
        }
    };
    auto unit = [&](Raw q) -> f2 {  // create_unit_float_image of the pair, then the clamped-column meaning of it
        f2 v;
        if constexpr (U8) v = f2{s_lut[q & 255u], s_lut[(q >> 8) & 255u]};
        else v = q;
        return fix_pair(v, C);
    };
    Raw q[R];
#pragma unroll
    for (int i = 0; i < PF; ++i) q[i] = feed(i);
    const f2 zero = {0.0f, 0.0f};
    f2 RH[P];
#pragma unroll
    for (int i = 0; i < P; ++i) RH[i] = zero;
    if (U8) __syncthreads();  // the table

    for (int t0 = 0; t0 < T_; t0 += P) {
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int t = t0 + k;  // (whole groups of P rows: the rows past T_ store nothing)
            const int v = v0 + t, c = v - S;
            q[(k + PF) % R] = feed(t + PF);
            float* const buf = &s_row[t & 1][0];
            const f2 own = unit(q[k % R]);
            *reinterpret_cast<f2*>(buf + wi) = own;
            __syncthreads();
            f2 a, b, m, d, e;
            if (!rare(edge_h)) {
                a = *reinterpret_cast<const f2*>(buf + wi - 2); b = lds2(buf + wi - 1); m = own;
                d = lds2(buf + wi + 1); e = *reinterpret_cast<const f2*>(buf + wi + 2);
            } else {
                const float* r0 = buf + i0;
                const float* r1 = buf + i1;
                a = f2{r0[0], r1[0]}; b = f2{r0[1], r1[1]}; m = f2{r0[2], r1[2]}; d = f2{r0[3], r1[3]}; e = f2{r0[4], r1[4]};
            }
            RH[k] = tap5(a, b, m, d, e, tp);
            const f2 o[1] = {tap5(RH[(k + 1) % P], RH[(k + 2) % P], RH[(k + 3) % P], RH[(k + 4) % P], RH[k], tp)};
            store_filled<S, 1, ODDW, false>(rout, C, w, h, c, t >= 2 * S && c >= cs && c < ce, o);
        }
    }
}

}  // namespace
namespace launch {
static int g_det_min_rows = 0, g_lvl_min_rows = 0, g_head_min_rows = 0;  // (akz_debug_set_schedule keys 7, 8: measurement; 0 = the planners' rules)
void march_min_band_rows(int detector, int level) {
    g_det_min_rows = detector;
    g_lvl_min_rows = level % 1000;
    g_head_min_rows = level / 1000;  // (key 8 = 1000 * rows of the level-0 marches + rows of the level marches)
}
}  // namespace launch
namespace {
using launch::g_det_min_rows;
using launch::g_lvl_min_rows;
using launch::g_head_min_rows;

inline MarchGrid plan_level_march(uint32_t w, uint32_t h, uint32_t n, dim3* grid, int fill_wg = 3, int min_band_rows = 64) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const int fill = fill_wg, min_rows = min_band_rows;  // (level kernels: 3 / 64, swept in round 2: flat within +-1.5 % around these)
    MarchGrid g;
    g.nstrips = (int)((w + USE - 1) / USE);
    const int rows = (int)h - 2;
    const long cols = (long)n * g.nstrips;
    const long want = ((long)cus * fill + cols - 1) / cols;
    long nb = std::max<long>(1, std::min<long>(want, std::max(1, rows / min_rows)));
    constexpr int edge_pct = 75;  // rows of an edge band in percent of a middle band's
    if (nb >= 3) {
        // 2 edge bands of rho * m rows + (nb - 2) middle bands of m rows = rows
        const double m = (double)rows / ((double)(nb - 2) + 2.0 * edge_pct / 100.0);
        g.band_rows = std::max(1, (int)(m + 0.999));
        g.edge_rows = std::max(1, (int)((rows - (long)(nb - 2) * g.band_rows) / 2));
        g.nbands = (int)nb;
        if (g.edge_rows + (long)(nb - 2) * g.band_rows >= rows) {  // degenerate: uniform bands
            g.band_rows = (int)((rows + nb - 1) / nb);
            g.edge_rows = g.band_rows;
            g.nbands = (rows + g.band_rows - 1) / g.band_rows;
        }
    } else {
        g.band_rows = (int)((rows + nb - 1) / nb);
        g.edge_rows = g.band_rows;
        g.nbands = (rows + g.band_rows - 1) / g.band_rows;
    }
    g.total = (int)(cols * g.nbands);
    *grid = dim3((unsigned)((g.total + 7) / 8 * 8));
    return g;
}

inline MarchGrid plan_march(uint32_t w, uint32_t h, uint32_t n, int S, dim3* grid, int fill_wg = 3, int min_band_rows = 64) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    // Bands: each one re-warms the rings (4S+3 extra rows), so they are as tall as the machine allows — enough
    // workgroups for `fill` resident workgroups per CU, never shorter than min_rows interior rows.
    const int fill = fill_wg, min_rows = min_band_rows;
    const int use = USE;
    (void)S;
    MarchGrid g;
    g.nstrips = (int)((w + use - 1) / use);
    const int rows = (int)h - 2 * S;
    const long cols = (long)n * g.nstrips;
    const long want = ((long)cus * fill + cols - 1) / cols;  // bands needed to fill the chip
    long nb = std::max<long>(1, std::min<long>(want, std::max(1, rows / min_rows)));
    g.band_rows = (int)((rows + nb - 1) / nb);
    g.edge_rows = g.band_rows;
    g.nbands = (rows + g.band_rows - 1) / g.band_rows;
    g.total = (int)(cols * g.nbands);
    *grid = dim3((unsigned)((g.total + 7) / 8 * 8));
    return g;
}

}  // namespace

namespace launch {

// the candidate rectangle of scale_space_extrema.rs:32-42 and :80-87 (the reference's float expressions; both border
// tests are monotone in the coordinate, so the admissible coordinates form one interval)
static void admissible_range(uint32_t dim, float border_m, int* lo, int* hi) {
    auto out_lo = [&](int v) { return (roundf((float)v - border_m) - 1.0f) < 0.0f; };
    auto out_hi = [&](int v) { return (roundf((float)v + border_m) + 1.0f) >= (float)dim; };
    int a = 1, b = (int)dim - 2;
    while (a <= b && out_lo(a)) ++a;
    while (b >= a && out_hi(b)) --b;
    *lo = a;
    *hi = b;  // empty when lo > hi
}

bool detector_march_supported(uint32_t sigma, uint32_t w, uint32_t h, float border_m, bool nms) {
    if (sigma < 1 || sigma > 4) return false;
    if (w < 4 * sigma + 8 || h < 4 * sigma + 8) return false;
    if ((uint64_t)w * h > (1ull << 28)) return false;  // buffer range checks: one image's plane stays below 2^30 bytes
    // the extrema test reads Ldet one pixel around a candidate: keep that ring inside the interior rows / columns
    return !nms || border_m >= (float)(sigma + 2);
}


template <int S, bool NMS, bool KEEP>
static void launch_detector_march(hipStream_t s, const float* lsmooth, float* lx, float* ly, float* lxx, float* lyy, float* lxy,
                                  float* ldet_out, uint32_t w, uint32_t h, uint32_t n, float kn, float kwn, float quat,
                                  const MarchNms& na) {
    dim3 gr;
#ifndef AKZ_DET_FILL
#define AKZ_DET_FILL 3
#endif
    // (bands of at least 40 interior rows: only a small job is cut that fine -- a lone 4K frame into 53 bands x 8 strips instead
    // of 33 x 8, one workgroup per compute unit and each marching 80 rows one after the other: 100 -> 70 us per launch; with 64
    // rows, the rule until round 6, a lone 4K call took 1.62 ms, with 40 or 32 1.55, with 24 or 16 1.60.  Batches are cut by the
    // fill target long before.)
#ifndef AKZ_DET_MINROWS
#define AKZ_DET_MINROWS 40
#endif
    // (... and 24 where the job is four strip columns or fewer -- a lone 1080p frame, a batch-path job since the end of round 6:
    // 0.613 -> 0.596 ms per call, 0.441 -> 0.416 per streamed frame; from five columns on 40 and 32 measure the same, 24 worse)
    const int few_cols_rows = (uint64_t)n * ((w + USE - 1) / USE) <= 4 ? 24 : AKZ_DET_MINROWS;
    const MarchGrid mg = plan_march(w, h, n, S, &gr, AKZ_DET_FILL, g_det_min_rows > 0 ? g_det_min_rows : few_cols_rows);
    if (w & 1u)
        hipLaunchKernelGGL((k_detector_march<S, NMS, KEEP, true>), gr, dim3(MT), 0, s, lsmooth, lx, ly, lxx, lyy, lxy, ldet_out,
                           (int)w, (int)h, mg, kn, kwn, quat, na);
    else
        hipLaunchKernelGGL((k_detector_march<S, NMS, KEEP, false>), gr, dim3(MT), 0, s, lsmooth, lx, ly, lxx, lyy, lxy, ldet_out,
                           (int)w, (int)h, mg, kn, kwn, quat, na);
}
#define AKZ_MARCH(S)                                                                                              \
    case S: {                                                                                                     \
        if (d_cand && keep)                                                                                       \
            launch_detector_march<S, true, true>(s, lsmooth, lx, ly, lxx, lyy, lxy, ldet_out, w, h, n, kn, kwn, quat, na);   \
        else if (d_cand)                                                                                          \
            launch_detector_march<S, true, false>(s, lsmooth, lx, ly, lxx, lyy, lxy, ldet_out, w, h, n, kn, kwn, quat, na);  \
        else if (keep)                                                                                            \
            launch_detector_march<S, false, true>(s, lsmooth, lx, ly, lxx, lyy, lxy, ldet_out, w, h, n, kn, kwn, quat, na);  \
        else                                                                                                      \
            launch_detector_march<S, false, false>(s, lsmooth, lx, ly, lxx, lyy, lxy, ldet_out, w, h, n, kn, kwn, quat, na); \
    } break;

// One level's detector response (+ extrema candidates when d_cand is given) in one launch of k_detector_march.
// lxx / lyy / lxy may be null together (the planes are then not written).
void detector_march(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx, float* lyy,
                    float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n, uint32_t level, float thr,
                    float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count) {
    const Taps m = taps_scharr_main(sigma);
    const float kn = m.wgt[0], kwn = m.wgt[1];
    const float quat = (float)(sigma * sigma * sigma * sigma);
    const bool keep = lxx && lyy && lxy;
    MarchNms na{level, thr, 0, -1, 0, -1, d_cand, cap, d_count};
    if (d_cand) {
        admissible_range(w, border_m, &na.xlo, &na.xhi);
        admissible_range(h, border_m, &na.ylo, &na.yhi);
    }
    switch (sigma) {
        AKZ_MARCH(1) AKZ_MARCH(2) AKZ_MARCH(3) AKZ_MARCH(4)
        default: break;
    }
}
#undef AKZ_MARCH

// 5-tap gaussian_blur as a march (u8 input: rows of even width are 2-byte aligned, which the 16-bit loads need)
bool blur5_march_supported(uint32_t w, uint32_t h, uint32_t ntaps) {
    return ntaps == 5 && w >= 16 && h >= 16 && (uint64_t)w * h <= (1ull << 28);
}
template <typename T>
static void blur5_march_t(hipStream_t s, const T* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k) {
    BlurTaps5 tp;
    for (int i = 0; i < 5; ++i) tp.k[i] = k[i];
    dim3 gr;
    // a light kernel (36-40 registers): six workgroups per compute unit instead of three, 86 -> 66 us per 32 x 1080p
    const MarchGrid mg = plan_march(w, h, n, 2, &gr, 6, g_head_min_rows > 0 ? g_head_min_rows : 32);
    if (w & 1u) hipLaunchKernelGGL((k_blur5_march<T, true>), gr, dim3(MT), 0, s, in, out, (int)w, (int)h, mg, tp);
    else hipLaunchKernelGGL((k_blur5_march<T, false>), gr, dim3(MT), 0, s, in, out, (int)w, (int)h, mg, tp);
}
void blur5_march_u8(hipStream_t s, const uint8_t* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k) {
    blur5_march_t<uint8_t>(s, in, out, w, h, n, k);
}
void blur5_march_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k) {
    blur5_march_t<float>(s, in, out, w, h, n, k);
}

// compute_contrast_factor's two passes over gaussian_blur(in, 1.0 with the 3 taps g3) as marches (d_hmax_bits / d_hist
// zeroed by the caller; d_thr: n x (nbins + 1) doubles, filled between the passes).  nbins <= 640.
bool contrast_march_supported(uint32_t w, uint32_t h, uint32_t ntaps, uint32_t nbins) {
    return ntaps == 3 && nbins >= 1 && nbins <= 640 && level_march_supported(w, h);
}
void contrast_march(hipStream_t s, const float* in, uint32_t w, uint32_t h, uint32_t n, const float* g3,
                    unsigned long long* d_hmax_bits, uint32_t nbins, uint32_t* d_hist, double* d_thr) {
    const Taps m = taps_scharr_main(1);
    const ContrastMarchArgs ca{d_hmax_bits, d_hist, nbins, d_thr};
    dim3 gr;
    // light kernels (52-59 registers, 8 to 20 KB of LDS): eight workgroups per compute unit hide the row latency that
    // three leave exposed (maximum pass 103 -> 82 us per 32 x 1080p); a band warms up over six rows only
    const MarchGrid mg = plan_level_march(w, h, n, &gr, 8, g_head_min_rows > 0 ? g_head_min_rows : 32);
    if (w & 1u)
        hipLaunchKernelGGL((k_contrast_march<1, true>), gr, dim3(MT), 0, s, in, (int)w, (int)h, mg, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], ca);
    else
        hipLaunchKernelGGL((k_contrast_march<1, false>), gr, dim3(MT), 0, s, in, (int)w, (int)h, mg, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], ca);
    contrast_thresholds(s, d_hmax_bits, nbins, n, d_thr);
    const size_t lds = ((size_t)(MT / 64) * CM_COPIES * nbins) * sizeof(unsigned) + (size_t)(nbins + 1) * sizeof(double);
    if (w & 1u)
        hipLaunchKernelGGL((k_contrast_march<2, true>), gr, dim3(MT), lds, s, in, (int)w, (int)h, mg, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], ca);
    else
        hipLaunchKernelGGL((k_contrast_march<2, false>), gr, dim3(MT), lds, s, in, (int)w, (int)h, mg, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], ca);
}

// (half-resolution launches of a 32-frame batch: with 64-row bands there are 512 workgroups for 768 places; 40-row bands
// fill them, 101 -> 94 us per launch)
#ifndef AKZ_LVL_FILL
#define AKZ_LVL_FILL 3
#endif
static int level_min_band_rows(uint32_t w, uint32_t h, uint32_t n) {
    if (g_lvl_min_rows > 0) return g_lvl_min_rows;
    // (below 48 Mpx the fill target decides, down to 20-row bands: a lone 4K frame's three full-resolution launches 76 -> ~60 us
    // each, 96 x 8 workgroups instead of 54 x 8; 20, 16, 12 and 8 rows measure the same)
    return (uint64_t)w * h * n < (48u << 20) ? 20 : 64;
}
// Test hook (CPU): the bands the planners cut an n-image batch of w x h into -- kind 0: detector / blur march with
// kernel half width S, kind 1: level march.  Writes up to cap [cs, ce) pairs, returns the number of bands.
uint32_t march_band_rows(int kind, uint32_t w, uint32_t h, uint32_t n, int S, int32_t* cs_ce, uint32_t cap) {
    dim3 gr;
    const MarchGrid g = kind == 1 ? plan_level_march(w, h, n, &gr, AKZ_LVL_FILL, level_min_band_rows(w, h, n)) : plan_march(w, h, n, S, &gr);
    for (int b = 0; b < g.nbands && (uint32_t)b < cap; ++b) {
        int cs, ce;
        if (kind == 1) level_band_rows(g, b, (int)h, &cs, &ce);
        else uniform_band_rows(g, b, S, (int)h, &cs, &ce);
        cs_ce[2 * b] = cs;
        cs_ce[2 * b + 1] = ce;
    }
    return (uint32_t)g.nbands;
}

bool level_march_supported(uint32_t w, uint32_t h) { return w >= 16 && h >= 16 && (uint64_t)w * h <= (1ull << 28); }

template <int NS, bool KEEPSTEP>
static void launch_level_march(hipStream_t s, dim3 gr, const MarchGrid& mg, const float* prev, float* lsmooth, float* lflow,
                               float* lt_out, float* lstep, uint32_t w, uint32_t h, const float* g3, const Taps& m,
                               const double* d_k, uint32_t k_pow, const LevelTaus& ht, uint32_t pw, uint32_t ph) {
    if constexpr (NS == 4) {  // (the first level of an octave always has four or more steps)
        if (pw) {
            hipLaunchKernelGGL((k_level_march<NS, KEEPSTEP, false, true>), gr, dim3(MT), 0, s, prev, lsmooth, lflow, lt_out, lstep, (int)w,
                               (int)h, mg, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow, ht, (int)pw, (int)ph);
            return;
        }
    }
    if (w & 1u)
        hipLaunchKernelGGL((k_level_march<NS, KEEPSTEP, true>), gr, dim3(MT), 0, s, prev, lsmooth, lflow, lt_out, lstep, (int)w,
                           (int)h, mg, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow, ht);
    else
        hipLaunchKernelGGL((k_level_march<NS, KEEPSTEP, false>), gr, dim3(MT), 0, s, prev, lsmooth, lflow, lt_out, lstep, (int)w,
                           (int)h, mg, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow, ht);
}
#define AKZ_LEVEL(NS)                                                                                                        \
    case NS: {                                                                                                               \
        if (lstep) launch_level_march<NS, true>(s, gr, mg, prev, lsmooth, lflow, lt_out, lstep, w, h, g3, m, d_k, k_pow, ht, pw, ph);  \
        else launch_level_march<NS, false>(s, gr, mg, prev, lsmooth, lflow, lt_out, lstep, w, h, g3, m, d_k, k_pow, ht, pw, ph);       \
    } break;

// Level preparation + the level's first n_steps (1..4) diffusion steps in one launch of k_level_march.  prev: the
// level's Lt before diffusion (the previous level's final Lt, or its 2x2 mean), must not alias lt_out.  lstep (may be
// null) receives the increment of the LAST fused step.
// the 2x2 mean of a pw x ph plane can be formed inside the level kernel (level_march with pw, ph given)
bool level_march_half_supported(uint32_t w, uint32_t h, uint32_t pw, uint32_t ph, uint32_t n_steps) {
    return n_steps == 4 && (pw & 3u) == 0 && (w & 1u) == 0 && w == pw / 2 && h == ph / 2 && (uint64_t)pw * ph <= (1ull << 28);
}
void level_march(hipStream_t s, const float* prev, float* lsmooth, float* lflow, float* lt_out, float* lstep, uint32_t w,
                 uint32_t h, uint32_t n, const float* g3, const double* d_k, uint32_t k_pow, const float* half_taus,
                 uint32_t n_steps, uint32_t pw, uint32_t ph) {
    const Taps m = taps_scharr_main(1);
    LevelTaus ht;
    for (uint32_t i = 0; i < 4; ++i) ht.half_tau[i] = i < n_steps ? half_taus[i] : 0.0f;
    dim3 gr;
    const MarchGrid mg = plan_level_march(w, h, n, &gr, AKZ_LVL_FILL, level_min_band_rows(w, h, n));
    switch (n_steps) {
        AKZ_LEVEL(1) AKZ_LEVEL(2) AKZ_LEVEL(3) AKZ_LEVEL(4)
        default: break;
    }
}
#undef AKZ_LEVEL

}  // namespace launch
}  // namespace akz
