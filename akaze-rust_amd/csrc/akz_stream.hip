// Barrier-free streaming kernels for gfx950: the level preparation (lib.rs:80-105) and the detector
// derivatives (detector_response.rs:8-55) as register-ring stencil chains.
//
// Every lane owns a column of four consecutive pixels and marches down the rows of a strip:
//
//   * a horizontal pass on a plane that is in HBM reads its taps (x-S, x, x+S) as dword-aligned 16-byte
//     global loads of the input row (a wave reads 1 KiB of a row per load, the shifted loads hit L1); a
//     horizontal pass on a row that exists only in registers (k_prep_stream) takes the neighbours of a
//     quad from the adjacent lanes with DPP wave shifts;
//   * the horizontal results of the last 2S+1 rows live in a REGISTER ring (the loop is unrolled by the
//     ring length, so every ring index is static); the vertical pass combines ring rows r-2S, r-S, r and
//     the outputs go straight to HBM as 16-byte stores;
//   * there is no LDS and no barrier; the next row's loads are issued before the current row's arithmetic.
//
// fill_border (types/image.rs:239-260) is reproduced exactly: a pass result at (x, y) is the interior
// result at (clamp(x,S,w-1-S), clamp(y,S,h-1-S)).  Columns: lanes whose four pixels are not all interior
// ("edge lanes") evaluate every pixel at its clamped column with scalar loads.  Rows: the H pass of row v
// reads input row clamp(v), and the wave that produces interior row S (h-1-S) also stores it to rows
// 0..S-1 (h-S..h-1).
//
// Work split: one wave per (image, row band, strip), strips fastest, so that the waves running side by
// side work on the same image rows; bands are sized to give every resident wave slot one wave (all waves
// finish together), but never shorter than a few ring lengths (each band re-warms its ring).
//
// Where they are used (akz_api.cpp picks per launch; both families give identical bytes): the streaming
// preparation kernel is ~2x faster than the LDS-tiled one for batches (4.2 vs 2.2 TB/s at 32 x 1080p); the
// streaming detector pair wins only when Lxx/Lyy/Lxy are not written out.  A 2 reads : 4 writes kernel
// tops out near 4.4-5.2 TB/s on MI355X even for a plain copy-like loop (tools/membw), so the detector
// kernels are bounded by their write mix, not by their structure.
//
// Arithmetic is the reference's: f32 mul then add, taps left to right starting from 0.0f, no FMA.
// The off-axis Scharr taps [-1, 0.., 0, ..0, 1] are evaluated as (0.0f - a) + c, which is
// bit-identical to ((0.0f + -1.0f*a) + 0.0f*b) + 1.0f*c for every finite b (the zero taps add +-0
// to an accumulator that is never -0).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "akz_internal.hpp"

namespace akz {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // four pixels of a row, dword-aligned

constexpr int WAVE = 64, SNT = 256;  // 4 independent waves per workgroup
#ifndef AKZ_STREAM_PF
#define AKZ_STREAM_PF 1
#endif
constexpr int PF = AKZ_STREAM_PF;     // rows of input taps in flight per wave

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ f4 tap_main(f4 a, f4 b, f4 c, float kn, float kwn) {
    return ((0.0f + kn * a) + kwn * b) + kn * c;
}
__device__ __forceinline__ f4 tap_off(f4 a, f4 c) { return (0.0f - a) + c; }

struct StreamGrid {
    int nstrips;    // strips per image row
    int nbands;     // row bands per image
    int band_rows;  // interior rows per band
    long waves;     // n * nbands * nstrips
};

// Column geometry of one lane inside its strip.  A quad that would cross the right image edge is
// shifted left to end at the last column (it then overlaps its neighbour and recomputes a few of its
// pixels with identical results), so every store is one 16-byte store.
struct Lane {
    int x;           // column of the lane's first pixel (halo lanes may lie outside the image)
    int first;       // first pixel of the quad this lane OWNS (reports candidates for); 4 = none
    bool edge;       // some pixel is evaluated at a clamped column
    bool store;      // this lane stores its quad
    unsigned cx[4];  // evaluation columns of the four pixels
};
template <int S, int HL>
__device__ __forceinline__ Lane make_lane(int strip, int lane, int w) {
    constexpr int SW = 4 * (WAVE - 2 * HL);
    Lane L;
    const int xn = strip * SW + 4 * (lane - HL);  // nominal position
    L.store = lane >= HL && lane < WAVE - HL && xn < w;
    L.x = L.store ? min(xn, w - 4) : xn;
    L.first = L.store ? xn - L.x : 4;
    L.edge = !(L.x >= S && L.x + 3 <= w - 1 - S);
#pragma unroll
    for (int i = 0; i < 4; ++i) L.cx[i] = (unsigned)clampi(L.x + i, S, w - 1 - S);
    return L;
}
// taps a = row[cx-S], b = row[cx], c = row[cx+S] of the lane's four pixels (edge lanes only)
template <int S, bool NEED_B>
__device__ __forceinline__ void fetch_edge(const float* __restrict__ row, const Lane& L, f4& a, f4& b, f4& c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = row[L.cx[i] - S];
        if (NEED_B) b[i] = row[L.cx[i]];
        c[i] = row[L.cx[i] + S];
    }
}
template <int S>
__device__ __forceinline__ void fetch_vec_x(const float* __restrict__ row, int x, f4& a, f4& b, f4& c) {
    a = *reinterpret_cast<const f4u*>(row + x - S);
    b = *reinterpret_cast<const f4u*>(row + x);
    c = *reinterpret_cast<const f4u*>(row + x + S);
}
template <int S>
__device__ __forceinline__ void fetch_edge_x(const float* __restrict__ row, const unsigned (&cx)[4], f4& a, f4& b, f4& c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = row[cx[i] - S];
        b[i] = row[cx[i]];
        c[i] = row[cx[i] + S];
    }
}
template <int S, bool NEED_B>
__device__ __forceinline__ void fetch_vec(const float* __restrict__ row, const Lane& L, f4& a, f4& b, f4& c) {
    a = *reinterpret_cast<const f4u*>(row + L.x - S);
    if (NEED_B) b = *reinterpret_cast<const f4u*>(row + L.x);
    c = *reinterpret_cast<const f4u*>(row + L.x + S);
}
// one output row of N planes; ro = offset of the row's first pixel
template <int N>
__device__ __forceinline__ void store_rows(float* const (&plane)[N], size_t ro, const Lane& L, const f4 (&v)[N]) {
    if (L.store) {
#pragma unroll
        for (int i = 0; i < N; ++i) plane_store4u(plane[i] + ro + L.x, v[i]);
    }
}
// interior row c of the output planes, plus (two rows per strip) the border rows that copy it
template <int S, int N>
__device__ __forceinline__ void store_filled(float* const (&plane)[N], const Lane& L, int w, int h, int c, const f4 (&v)[N]) {
    store_rows<N>(plane, (size_t)c * w, L, v);
    if (c == S || c == h - 1 - S) {  // wave-uniform
        if (c == S) {
#pragma nounroll
            for (int y = 0; y < S; ++y) store_rows<N>(plane, (size_t)y * w, L, v);
        }
        if (c == h - 1 - S) {
#pragma nounroll
            for (int y = h - S; y < h; ++y) store_rows<N>(plane, (size_t)y * w, L, v);
        }
    }
}

// One wave per (image, band, strip), strips fastest: the waves that run side by side read and write the
// same rows of the image, i.e. the same DRAM pages and TLB entries.  Everything here is wave-uniform and
// stays in scalar registers.
struct Piece {
    int img, strip, cs, ce;  // interior rows [cs, ce), S <= cs, ce <= h-S; empty when cs >= ce
};
__device__ __forceinline__ Piece wave_piece(long wave, const StreamGrid& g, int S, int h) {
    Piece p;
    const long per = (long)g.nbands * g.nstrips;
    p.img = (int)(wave / per);
    const int rem = (int)(wave - (long)p.img * per);
    const int band = rem / g.nstrips;
    p.strip = rem - band * g.nstrips;
    p.cs = S + band * g.band_rows;
    p.ce = min(p.cs + g.band_rows, h - S);
    return p;
}
// the wave index as a scalar: everything derived from it (pieces, row pointers, loop bounds) is then
// provably uniform and the compiler keeps it in SGPRs with scalar branches
__device__ __forceinline__ long wave_index() {
    return (long)blockIdx.x * (SNT / WAVE) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// ---------------------------------------------------------------------------------------------
// Multiscale first derivatives (detector_response.rs:9-10): Lx = V_off(H_main(Ls)), Ly = V_main(H_off(Ls))
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(SNT, (S <= 2 ? 4 : S == 3 ? 3 : 2))  // second argument: waves per SIMD the register budget must allow
k_deriv1_stream(const float* __restrict__ ls, float* __restrict__ lx_out, float* __restrict__ ly_out, int w, int h,
                StreamGrid g, float kn, float kwn) {
    constexpr int P = 2 * S + 1;
    const int lane = threadIdx.x & (WAVE - 1);
    const long wave = wave_index();
    if (wave >= g.waves) return;
    {
        const Piece pc = wave_piece(wave, g, S, h);
        if (pc.cs >= pc.ce) return;
        const Lane L = make_lane<S, 0>(pc.strip, lane, w);
        const size_t base = (size_t)pc.img * (size_t)w * (size_t)h;
        const float* in = ls + base;
        float* const out[2] = {lx_out + base, ly_out + base};
        const int v0 = pc.cs - S, T = (pc.ce - pc.cs) + 2 * S;  // H rows v0 .. v0+T-1
        f4 rM[P], rO[P];
        f4 qa[PF], qb[PF], qc[PF];  // taps of rows t .. t+PF-1, in flight ahead of the arithmetic
        auto fetch = [&](int t, f4& fa, f4& fb, f4& fc) {
            const float* row = in + (size_t)clampi(v0 + std::min(t, T - 1), S, h - 1 - S) * w;
            if (!L.edge) fetch_vec<S, true>(row, L, fa, fb, fc);
            else fetch_edge<S, true>(row, L, fa, fb, fc);
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) fetch(i, qa[i], qb[i], qc[i]);
        for (int t0 = 0; t0 < T; t0 += P) {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int t = t0 + k;
                if (t < T) {
                    f4 na, nb, nc;
                    fetch(t + PF, na, nb, nc);
                    const f4 a = qa[0], b = qb[0], c = qc[0];
                    rM[k] = tap_main(a, b, c, kn, kwn);
                    rO[k] = tap_off(a, c);
                    if (t >= 2 * S) {
                        const int k0 = (k + 1) % P, k1 = (k + P - S) % P;  // rows c-S, c; k = row c+S
                        const f4 v[2] = {tap_off(rM[k0], rM[k]), tap_main(rO[k0], rO[k1], rO[k], kn, kwn)};
                        store_filled<S, 2>(out, L, w, h, v0 + t - S, v);
                    }
#pragma unroll
                    for (int i = 0; i + 1 < PF; ++i) { qa[i] = qa[i + 1]; qb[i] = qb[i + 1]; qc[i] = qc[i + 1]; }
                    qa[PF - 1] = na; qb[PF - 1] = nb; qc[PF - 1] = nc;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Second derivatives, Hessian determinant and (NMS) the extrema test of scale_space_extrema.rs:32-42,
// :80-87: Lxx = V_off(H_main(Lx)), Lyy = V_main(H_off(Ly)), Lxy = V_main(H_off(Lx)),
// Ldet = ((Lxx*Lyy) - (Lxy*Lxy)) * sigma^4.  A pixel is a candidate if Ldet > threshold, Ldet is
// strictly above its 4 neighbours and the descriptor window fits in the image; the host turns the
// last test into the rectangle [xlo,xhi] x [ylo,yhi] (same float expressions) and checks that it
// keeps candidates at least S+2 pixels away from every edge, so only interior rows and columns are
// ever tested.  With NMS one lane on each side of the wave is a halo lane: it computes Ldet for its
// neighbour but neither stores nor reports.
// ---------------------------------------------------------------------------------------------
struct StreamNms {
    unsigned level;
    float thr;
    int xlo, xhi, ylo, yhi;
    Candidate* cand;
    unsigned cap;
    unsigned* count;
};

// Extrema of a wave are collected in a small LDS buffer of that wave and appended to the global list in blocks: one
// device-scope atomic per CAND_BUF candidates instead of one each (a 32-frame batch emits 1.6 x 10^5 candidates, and
// that many atomics on one address cost 0.6 ms).  All helpers are called by the whole wave (wave-uniform control flow).
constexpr int CAND_BUF = 32;
__device__ __forceinline__ void wave_cands_flush(Candidate* buf, unsigned& n, const StreamNms& nms, int lane) {
    if (n == 0) return;
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(nms.count, n);
    base = (unsigned)__builtin_amdgcn_readfirstlane((int)base);
    const uint4* src = reinterpret_cast<const uint4*>(buf);
    uint4* dst = reinterpret_cast<uint4*>(nms.cand);
    for (unsigned e = (unsigned)lane; e < 2u * n; e += WAVE)  // 16-byte halves of the 32-byte records
        if (base + (e >> 1) < nms.cap) dst[2 * (size_t)base + e] = src[e];
    __builtin_amdgcn_wave_barrier();
    n = 0;
}
// every lane offers the pixels of its quad whose bit is set in m (at most two); `make(i)` builds the record of pixel i
template <typename F>
__device__ __forceinline__ void wave_cands_push(Candidate* buf, unsigned& n, unsigned m, const StreamNms& nms, int lane,
                                                F&& make) {
    while (__ballot(m != 0u)) {  // rare
        const bool have = m != 0u;
        const int i = have ? __ffs(m) - 1 : 0;
        m &= m - 1u;
        const unsigned long long b = __ballot(have);
        const unsigned nb = (unsigned)__popcll(b);  // <= 64 > CAND_BUF is possible: flush first, then at most 32 per round
        if (n + nb > (unsigned)CAND_BUF) wave_cands_flush(buf, n, nms, lane);
        const unsigned before = (unsigned)__popcll(b & ((1ull << lane) - 1ull));
        if (nb <= (unsigned)CAND_BUF) {
            if (have) buf[n + before] = make(i);
            n += nb;
        } else {  // more hits in one row of the wave than the buffer holds: two halves
            const bool lo = before < (unsigned)CAND_BUF;
            if (have && lo) buf[before] = make(i);
            n = min(nb, (unsigned)CAND_BUF);
            __builtin_amdgcn_wave_barrier();
            wave_cands_flush(buf, n, nms, lane);
            if (have && !lo) buf[before - CAND_BUF] = make(i);
            n = nb - (unsigned)CAND_BUF;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int S, bool NMS, bool KEEP>
__global__ void __launch_bounds__(SNT, (S == 1 ? 3 : 2))
k_deriv2_stream(const float* __restrict__ lx_in, const float* __restrict__ ly_in, float* __restrict__ lxx_out,
                float* __restrict__ lyy_out, float* __restrict__ lxy_out, float* __restrict__ ldet_out, int w, int h,
                StreamGrid g, float kn, float kwn, float quat, StreamNms nms) {
    constexpr int P = 2 * S + 1, HL = NMS ? 1 : 0, NOUT = KEEP ? 4 : 1;
    __shared__ Candidate s_cands[NMS ? SNT / WAVE : 1][NMS ? CAND_BUF : 1];  // per-wave extrema buffer
    Candidate* const cbuf = s_cands[NMS ? (threadIdx.x >> 6) : 0];
    unsigned cnum = 0;
    const int lane = threadIdx.x & (WAVE - 1);
    const long wave = wave_index();
    if (wave >= g.waves) return;
    {
        const Piece pc = wave_piece(wave, g, S, h);
        if (pc.cs >= pc.ce) return;
        const Lane L = make_lane<S, HL>(pc.strip, lane, w);
        const size_t base = (size_t)pc.img * (size_t)w * (size_t)h;
        const float* inx = lx_in + base;
        const float* iny = ly_in + base;
        float* out[NOUT];
        out[0] = ldet_out + base;
        if (KEEP) {
            out[1] = lxx_out + base;
            out[2] = lyy_out + base;
            out[3] = lxy_out + base;
        }
        float* const(&outc)[NOUT] = out;
        // extrema test: rows [cs, ce) of this piece; it needs Ldet of rows cs-1 and ce as well
        const int cb = NMS ? std::max(pc.cs - 1, S) : pc.cs;
        const int cl = NMS ? std::min(pc.ce + 1, h - S) : pc.ce;
        const int v0 = cb - S, T = (cl - cb) + 2 * S;
        unsigned xok = 0;  // bit i: pixel i of this lane may be a candidate
        if (NMS) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i >= L.first && L.x + i >= nms.xlo && L.x + i <= nms.xhi) xok |= 1u << i;
        }
        f4 rA[P], rB[P], rC[P];
        f4 dm2 = 0.0f, dm1 = 0.0f;  // Ldet of rows c-2, c-1
        f4 qxa[PF], qxb[PF], qxc[PF], qya[PF], qyc[PF];  // taps of rows t .. t+PF-1, in flight ahead of the arithmetic
        auto fetch = [&](int t, f4& fxa, f4& fxb, f4& fxc, f4& fya, f4& fyc) {
            const size_t ro = (size_t)clampi(v0 + std::min(t, T - 1), S, h - 1 - S) * w;
            f4 unused;
            if (!L.edge) {
                fetch_vec<S, true>(inx + ro, L, fxa, fxb, fxc);
                fetch_vec<S, false>(iny + ro, L, fya, unused, fyc);
            } else {
                fetch_edge<S, true>(inx + ro, L, fxa, fxb, fxc);
                fetch_edge<S, false>(iny + ro, L, fya, unused, fyc);
            }
        };
#pragma unroll
        for (int i = 0; i < PF; ++i) fetch(i, qxa[i], qxb[i], qxc[i], qya[i], qyc[i]);
        for (int t0 = 0; t0 < T; t0 += P) {
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const int t = t0 + k;
                if (t < T) {
                    f4 nxa, nxb, nxc, nya, nyc;
                    fetch(t + PF, nxa, nxb, nxc, nya, nyc);
                    const f4 xa = qxa[0], xb = qxb[0], xc = qxc[0], ya = qya[0], yc = qyc[0];
                    rA[k] = tap_main(xa, xb, xc, kn, kwn);  // H_main(Lx)
                    rB[k] = tap_off(ya, yc);                // H_off(Ly)
                    rC[k] = tap_off(xa, xc);                // H_off(Lx)
                    if (t >= 2 * S) {
                        const int k0 = (k + 1) % P, k1 = (k + P - S) % P;
                        const int cr = v0 + t - S;
                        f4 o[NOUT];
                        const f4 lxx = tap_off(rA[k0], rA[k]);
                        const f4 lyy = tap_main(rB[k0], rB[k1], rB[k], kn, kwn);
                        const f4 lxy = tap_main(rC[k0], rC[k1], rC[k], kn, kwn);
                        const f4 det = ((lxx * lyy) - (lxy * lxy)) * quat;
                        o[0] = det;
                        if (KEEP) {
                            o[1] = lxx;
                            o[2] = lyy;
                            o[3] = lxy;
                        }
                        const f4(&oc)[NOUT] = o;
                        if (cr >= pc.cs && cr < pc.ce) store_filled<S, NOUT>(outc, L, w, h, cr, oc);
                        if (NMS) {
                            const int y = cr - 1;  // row under test: dm1, between dm2 (above) and det (below)
                            if (t >= 2 * S + 2 && y >= pc.cs && y < pc.ce && y >= nms.ylo && y <= nms.yhi) {  // uniform
                                const float left = __shfl_up(dm1[3], 1), right = __shfl_down(dm1[0], 1);
                                const float xm[4] = {left, dm1[0], dm1[1], dm1[2]};
                                const float xp[4] = {dm1[1], dm1[2], dm1[3], right};
                                unsigned m = 0;
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float v = dm1[i];
                                    const bool hit = (v > nms.thr) & (v > xp[i]) & (v > xm[i]) & (v > dm2[i]) & (v > det[i]);
                                    m |= hit ? 1u << i : 0u;
                                }
                                m &= xok;
                                wave_cands_push(cbuf, cnum, m, nms, lane, [&](int i) {  // at most two pixels of a quad can be strict maxima
                                    Candidate cd;
                                    cd.level = nms.level;
                                    cd.idx = (unsigned)(y * w + L.x + i);
                                    cd.v = i == 0 ? dm1[0] : i == 1 ? dm1[1] : i == 2 ? dm1[2] : dm1[3];
                                    cd.xp = i == 0 ? dm1[1] : i == 1 ? dm1[2] : i == 2 ? dm1[3] : right;
                                    cd.xm = i == 0 ? left : i == 1 ? dm1[0] : i == 2 ? dm1[1] : dm1[2];
                                    cd.yp = i == 0 ? det[0] : i == 1 ? det[1] : i == 2 ? det[2] : det[3];
                                    cd.ym = i == 0 ? dm2[0] : i == 1 ? dm2[1] : i == 2 ? dm2[2] : dm2[3];
                                    cd.img = (unsigned)pc.img;
                                    return cd;
                                });
                            }
                            dm2 = dm1;
                            dm1 = det;
                        }
                    }
#pragma unroll
                    for (int i = 0; i + 1 < PF; ++i) {
                        qxa[i] = qxa[i + 1]; qxb[i] = qxb[i + 1]; qxc[i] = qxc[i + 1];
                        qya[i] = qya[i + 1]; qyc[i] = qyc[i + 1];
                    }
                    qxa[PF - 1] = nxa; qxb[PF - 1] = nxb; qxc[PF - 1] = nxc; qya[PF - 1] = nya; qyc[PF - 1] = nyc;
                }
            }
        }
    }
    if (NMS) wave_cands_flush(cbuf, cnum, nms, lane);
}

// ---------------------------------------------------------------------------------------------
// One level's preparation (lib.rs:80-105), streaming: Lt_i (the previous level's Lt, or its 2x2 mean,
// types/image.rs:102-118) -> Lsmooth_i = gaussian_blur(Lt_i, 1.0) -> scale-1 Scharr pair -> Lflow_i =
// pm_g2.  Four passes with half width 1, chained per row in registers:
//
//   input row v --H_g--> G ring (3 rows) --V_g--> Lsmooth row u=v-1 --H_scharr--> HM/HO rings (3 rows)
//               --V_scharr--> (Lx1, Ly1) of row c=v-2 --pm_g2--> Lflow row c
//
// The second horizontal pass works on a row that exists only in registers: the x-1 / x+1 neighbours of
// a quad come from the adjacent lanes (DPP wave shifts), so one lane on each side of the wave is a halo
// lane.  fill_border: columns as in the kernels above (stage 1 evaluates edge pixels at their clamped
// column, so every lane holds the FILLED Lsmooth of its own columns; stage 2 copies the results of
// columns 1 / w-2 to columns 0 / w-1); rows: Lsmooth row 0 (h-1) is row 1 (h-2), which the first (last)
// band handles by entering the scharr ring slot of that row twice.
// ---------------------------------------------------------------------------------------------
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

__device__ __forceinline__ float from_left_lane(float v) {  // lane i receives lane i-1 (DPP wave_shr:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_right_lane(float v) {  // lane i receives lane i+1 (DPP wave_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ f4 tap3(f4 a, f4 b, f4 c, float k0, float k1, float k2) {
    return ((0.0f + k0 * a) + k1 * b) + k2 * c;
}
__device__ __forceinline__ float pm_g2_px(float lx, float ly, double inverse_k) {  // lib.rs:30-37
    const double dx = (double)lx, dy = (double)ly;
    return (float)(1.0 / (1.0 + inverse_k * (dx * dx + dy * dy)));
}
__device__ __forceinline__ float mean4(float p00, float p01, float p10, float p11) {  // image.rs:108-114
    float v = 0.0f;
    v = v + p00;  // (2x,   2y)
    v = v + p01;  // (2x,   2y+1)
    v = v + p10;  // (2x+1, 2y)
    v = v + p11;  // (2x+1, 2y+1)
    return v / 4.0f;
}

struct PrepLane {
    int x;
    bool edge;       // some pixel is evaluated at a clamped column (or lies outside the image)
    bool vst, sst;   // stores its quad with one 16-byte store / pixel by pixel (quad crosses the right edge)
    bool fix0;       // holds column 0
    int i0;          // component that holds column w-1, or -1
    unsigned cx[4];  // evaluation columns
};
__device__ __forceinline__ PrepLane make_prep_lane(int strip, int lane, int w) {
    PrepLane L;
    L.x = strip * (4 * (WAVE - 2)) + 4 * (lane - 1);
    L.edge = !(L.x >= 1 && L.x + 3 <= w - 2);
    const bool owner = lane >= 1 && lane < WAVE - 1 && L.x < w;
    L.vst = owner && L.x + 3 < w;
    L.sst = owner && !L.vst;
    L.fix0 = L.x == 0;
    L.i0 = (w - 1 >= L.x && w - 1 <= L.x + 3) ? w - 1 - L.x : -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) L.cx[i] = (unsigned)clampi(L.x + i, 1, w - 2);
    return L;
}
__device__ __forceinline__ void prep_store(float* __restrict__ row, const PrepLane& L, int w, f4 v) {
    if (L.vst) {
        plane_store4u(row + L.x, v);
    } else if (L.sst) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (L.x + i < w) plane_store(row + L.x + i, v[i]);
    }
}
// row y of a filled plane; rows 1 and h-2 are also the border rows 0 and h-1
__device__ __forceinline__ void prep_store_filled(float* __restrict__ plane, const PrepLane& L, int w, int h, int y, f4 v) {
    prep_store(plane + (size_t)y * w, L, w, v);
    if (y == 1) prep_store(plane, L, w, v);
    if (y == h - 2) prep_store(plane + (size_t)(h - 1) * w, L, w, v);
}

// Stage-1 taps of level row y (0 <= y < h): a, b, c = Lt_i at columns cx-1, cx, cx+1; raw = Lt_i at the lane's
// own columns (HALF only: that is the half-size image the level starts from).
template <bool HALF>
__device__ __forceinline__ void prep_fetch(const float* __restrict__ src, int pw, int w, int y, const PrepLane& L, f4& a,
                                           f4& b, f4& c, f4& raw) {
    if (!HALF) {
        const float* row = src + (size_t)y * w;
        if (!L.edge) {
            const f4 q0 = *reinterpret_cast<const f4u*>(row + L.x - 1);
            const f2u q1 = *reinterpret_cast<const f2u*>(row + L.x + 3);
            a = q0;
            b = f4{q0[1], q0[2], q0[3], q1[0]};
            c = f4{q0[2], q0[3], q1[0], q1[1]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = row[L.cx[i] - 1];
                b[i] = row[L.cx[i]];
                c[i] = row[L.cx[i] + 1];
            }
        }
        raw = b;
    } else {
        const float* r0 = src + (size_t)(2 * y) * pw;
        const float* r1 = r0 + pw;
        if (!L.edge) {
            const float* p0 = r0 + 2 * L.x - 2;
            const float* p1 = r1 + 2 * L.x - 2;
            const f4 u0 = *reinterpret_cast<const f4u*>(p0), u1 = *reinterpret_cast<const f4u*>(p0 + 4),
                     u2 = *reinterpret_cast<const f4u*>(p0 + 8);
            const f4 d0 = *reinterpret_cast<const f4u*>(p1), d1 = *reinterpret_cast<const f4u*>(p1 + 4),
                     d2 = *reinterpret_cast<const f4u*>(p1 + 8);
            const float m0 = mean4(u0[0], d0[0], u0[1], d0[1]), m1 = mean4(u0[2], d0[2], u0[3], d0[3]);
            const float m2 = mean4(u1[0], d1[0], u1[1], d1[1]), m3 = mean4(u1[2], d1[2], u1[3], d1[3]);
            const float m4 = mean4(u2[0], d2[0], u2[1], d2[1]), m5 = mean4(u2[2], d2[2], u2[3], d2[3]);
            a = f4{m0, m1, m2, m3};
            b = f4{m1, m2, m3, m4};
            c = f4{m2, m3, m4, m5};
            raw = b;
        } else {
            auto mean_at = [&](unsigned col) { return mean4(r0[2 * col], r1[2 * col], r0[2 * col + 1], r1[2 * col + 1]); };
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = mean_at(L.cx[i] - 1);
                b[i] = mean_at(L.cx[i]);
                c[i] = mean_at(L.cx[i] + 1);
                raw[i] = mean_at((unsigned)clampi(L.x + i, 0, w - 1));
            }
        }
    }
}

// MODE 0: level preparation.  MODE 1 / 2: the two passes of compute_contrast_factor (contrast_factor.rs:18-71) on
// the same chain — gaussian_blur(Lt0, 1.0), scale-1 Scharr pair, then over the interior pixels the maximum of
// sqrt(Lx^2 + Ly^2) in f64 (MODE 1: one atomicMax per wave) or its histogram (MODE 2: per-wave LDS histograms,
// flushed once) — without writing any plane.
struct ContrastArgs {
    unsigned long long* hmax_bits;  // per image, non-negative f64 as its bit pattern (orders like the value)
    unsigned* hist;                 // per image, nbins counters
    unsigned nbins;
};
constexpr int CHIST_COPIES = 4;  // sub-histograms per wave (lanes spread over them) against same-bin conflicts

template <bool HALF, int MODE>
__global__ void __launch_bounds__(SNT, 3)
k_prep_stream(const float* __restrict__ prev, float* __restrict__ lt_out, float* __restrict__ lsmooth,
              float* __restrict__ lflow, int w, int h, int pw, int ph, StreamGrid g, float g0, float g1, float g2,
              float kn, float kwn, const double* __restrict__ d_k, unsigned k_pow, ContrastArgs ca) {
    static_assert(!(HALF && MODE != 0), "the contrast passes read the level itself");
    extern __shared__ unsigned s_chist[];  // MODE 2: [wave][copy][bin]
    const int lane = threadIdx.x & (WAVE - 1);
    const long wave = wave_index();
    if (wave >= g.waves) return;
    const Piece pc = wave_piece(wave, g, 1, h);
    if (pc.cs >= pc.ce) return;
    const PrepLane L = make_prep_lane(pc.strip, lane, w);
    const bool edge_strip = pc.strip == 0 || pc.strip == g.nstrips - 1;
    const float* src = prev + (size_t)pc.img * (size_t)pw * (size_t)ph;
    const size_t base = (size_t)pc.img * (size_t)w * (size_t)h;
    float* ltp = HALF ? lt_out + base : nullptr;
    float* lsp = MODE == 0 ? lsmooth + base : nullptr;
    float* lfp = MODE == 0 ? lflow + base : nullptr;
    double inverse_k = 0.0, hmax = 0.0, gmax = 0.0;
    unsigned* myhist = nullptr;
    unsigned colmask = 0;  // MODE 1/2: pixels of this lane inside the interior columns 1..w-2 (contrast_factor.rs:35)
    if (MODE == 0) {
        double kc = d_k[pc.img];
        for (unsigned i = 0; i < k_pow; ++i) kc = kc * 0.75;  // lib.rs:84, one octave at a time in f64
        inverse_k = 1.0 / (kc * kc);
    } else {
        if (L.vst || L.sst)
            for (int i = 0; i < 4; ++i)
                if (L.x + i >= 1 && L.x + i <= w - 2) colmask |= 1u << i;
        if (MODE == 2) {
            hmax = __longlong_as_double((long long)ca.hmax_bits[pc.img]);
            unsigned* wh = s_chist + (size_t)(threadIdx.x >> 6) * CHIST_COPIES * ca.nbins;
            for (unsigned b = lane; b < CHIST_COPIES * ca.nbins; b += WAVE) wh[b] = 0u;  // wave-private: no barrier
            myhist = wh + (lane & (CHIST_COPIES - 1)) * ca.nbins;
        }
    }
    const int v0 = pc.cs - 2, T = (pc.ce - pc.cs) + 4;  // input rows v0 .. v0+T-1 (clamped to 1..h-2 when loaded)
    f4 a, b, c, raw;
    if (HALF) {  // rows 0 and h-1 of the half-size image are not on the filter path (their taps are clamped away)
        if (pc.cs == 1) {
            prep_fetch<true>(src, pw, w, 0, L, a, b, c, raw);
            prep_store(ltp, L, w, raw);
        }
        if (pc.ce == h - 1) {
            prep_fetch<true>(src, pw, w, h - 1, L, a, b, c, raw);
            prep_store(ltp + (size_t)(h - 1) * w, L, w, raw);
        }
    }
    f4 G[3], HM[3], HO[3];
    prep_fetch<HALF>(src, pw, w, clampi(v0, 1, h - 2), L, a, b, c, raw);
    for (int t0 = 0; t0 < T; t0 += 3) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int t = t0 + k;
            if (t < T) {
                const int km1 = (k + 2) % 3, km2 = (k + 1) % 3;
                const int v = v0 + t;
                f4 na, nb, nc, nraw;  // the next row's taps are in flight during this row's arithmetic
                prep_fetch<HALF>(src, pw, w, clampi(v0 + std::min(t + 1, T - 1), 1, h - 2), L, na, nb, nc, nraw);
                if (HALF && v >= pc.cs && v < pc.ce) prep_store(ltp + (size_t)v * w, L, w, raw);
                G[k] = tap3(a, b, c, g0, g1, g2);
                if (t >= 2) {
                    const int u = v - 1;
                    const f4 ls = tap3(G[km2], G[km1], G[k], g0, g1, g2);  // filled Lsmooth of the lane's columns
                    if (MODE == 0 && u >= pc.cs && u < pc.ce) prep_store_filled(lsp, L, w, h, u, ls);
                    const f4 la = f4{from_left_lane(ls[3]), ls[0], ls[1], ls[2]};
                    const f4 lc = f4{ls[1], ls[2], ls[3], from_right_lane(ls[0])};
                    f4 hm = tap_main(la, ls, lc, kn, kwn);
                    f4 ho = tap_off(la, lc);
                    if (MODE == 0 && edge_strip) {  // wave-uniform: results of columns 1 / w-2 are also those of columns 0 / w-1
                        const float ml = from_left_lane(hm[3]), ol = from_left_lane(ho[3]);
                        const f4 rm = hm, ro = ho;
                        if (L.fix0) { hm[0] = rm[1]; ho[0] = ro[1]; }
                        if (L.i0 == 0) { hm[0] = ml; ho[0] = ol; }
                        if (L.i0 == 1) { hm[1] = rm[0]; ho[1] = ro[0]; }
                        if (L.i0 == 2) { hm[2] = rm[1]; ho[2] = ro[1]; }
                        if (L.i0 == 3) { hm[3] = rm[2]; ho[3] = ro[2]; }
                    }
                    HM[k] = hm;
                    HO[k] = ho;
                    if (u == 1) { HM[km1] = hm; HO[km1] = ho; }              // Lsmooth row 0 is row 1
                    if (u == h - 1) { HM[k] = HM[km1]; HO[k] = HO[km1]; }    // Lsmooth row h-1 is row h-2
                    if (t >= 4) {
                        const int cr = u - 1;
                        const f4 lx1 = tap_off(HM[km2], HM[k]);
                        const f4 ly1 = tap_main(HO[km2], HO[km1], HO[k], kn, kwn);
                        if (MODE == 0) {
                            f4 fl;
#pragma unroll
                            for (int i = 0; i < 4; ++i) fl[i] = pm_g2_px(lx1[i], ly1[i], inverse_k);
                            if (cr >= pc.cs && cr < pc.ce) prep_store_filled(lfp, L, w, h, cr, fl);
                        } else if (cr >= pc.cs && cr < pc.ce) {  // interior rows 1..h-2 are exactly the band rows
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (!(colmask & (1u << i))) continue;
                                const double dx = (double)lx1[i], dy = (double)ly1[i];
                                const double ss = dx * dx + dy * dy;
                                if (MODE == 1) {  // sqrt is monotone and correctly rounded: max sqrt(s) == sqrt(max s)
                                    if (ss > gmax) gmax = ss;
                                    continue;
                                }
                                const double gm = sqrt(ss);
                                if (gm != 0.0) {
                                    const double f = floor((double)ca.nbins * (gm / hmax));
                                    const unsigned b = f >= (double)ca.nbins ? ca.nbins - 1u : (f > 0.0 ? (unsigned)f : 0u);
                                    atomicAdd(&myhist[b], 1u);
                                }
                            }
                        }
                    }
                }
                a = na; b = nb; c = nc; raw = nraw;
            }
        }
    }
    if (MODE == 1) {
        unsigned long long bits = (unsigned long long)__double_as_longlong(sqrt(gmax));
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(bits, o, 64);
            bits = other > bits ? other : bits;
        }
        if (lane == 0 && bits != 0ull) atomicMax(ca.hmax_bits + pc.img, bits);
    }
    if (MODE == 2) {
        const unsigned* wh = s_chist + (size_t)(threadIdx.x >> 6) * CHIST_COPIES * ca.nbins;
        for (unsigned b = lane; b < ca.nbins; b += WAVE) {
            unsigned v = 0;
            for (int cpy = 0; cpy < CHIST_COPIES; ++cpy) v += wh[cpy * ca.nbins + b];
            if (v) atomicAdd(&ca.hist[(size_t)pc.img * ca.nbins + b], v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused detector (detector_response.rs:8-55 + the extrema test): Lsmooth -> Lx, Ly -> Lxx, Lyy, Lxy, Ldet
// in ONE pass over the level: 4 B read and 24 B written per pixel instead of 12 + 24 for the two-kernel
// form (Lx and Ly are not read back).  Per input row v:
//
//   Lsmooth row v --H--> (Hm, Ho) ring in LDS (wave-private, 2S+1 rows) --V--> Lx, Ly of row u = v-S (stored)
//        --H, neighbours by DPP--> (A, B, C) register rings (2S+1 rows) --V--> Lxx, Lyy, Lxy, Ldet of row
//        c = v-2S (stored) --> extrema test of row c-1
//
// Columns: stage 1 evaluates edge pixels at their clamped column, so every lane holds the FILLED Lx / Ly
// of its own columns; the stage-2 H results of columns S / w-1-S are broadcast (readlane) into the
// columns left / right of them.  Rows: the filled Lx rows 0..S-1 (h-S..h-1) are row S (h-1-S): the first
// band enters that row's stage-2 H result into S+1 ring slots, the last band does the same for the
// bottom rows and then drains the S output rows that only need those copies.  Two halo lanes on each
// side of the wave (stage 2 and the extrema test each look one lane sideways): strips are 240 pixels.
// ---------------------------------------------------------------------------------------------
template <int I, int S>
__device__ __forceinline__ float col_minus(const f4& v) {  // value of column (own column I) - S
    if constexpr (I >= S) return v[I - S];
    else return from_left_lane(v[4 + I - S]);
}
template <int I, int S>
__device__ __forceinline__ float col_plus(const f4& v) {  // value of column (own column I) + S
    if constexpr (I + S <= 3) return v[I + S];
    else return from_right_lane(v[I + S - 4]);
}
template <int S>
__device__ __forceinline__ void shifted(const f4& v, f4& a, f4& c) {
    a = f4{col_minus<0, S>(v), col_minus<1, S>(v), col_minus<2, S>(v), col_minus<3, S>(v)};
    c = f4{col_plus<0, S>(v), col_plus<1, S>(v), col_plus<2, S>(v), col_plus<3, S>(v)};
}

struct DetLane {
    int x;
    bool edge;       // stage 1: some pixel is evaluated at a clamped column (or lies outside the image)
    bool vst, sst;   // owner: one 16-byte store / pixel by pixel (quad crosses the right edge)
    unsigned lo, hi; // bit i: column x+i < S  /  > w-1-S  (stage-2 H result is a copy of column S / w-1-S)
    unsigned cx[4];
};
template <int S>
__device__ __forceinline__ DetLane make_det_lane(int strip, int lane, int w) {
    DetLane L;
    L.x = strip * (4 * (WAVE - 4)) + 4 * (lane - 2);
    L.edge = !(L.x >= S && L.x + 3 <= w - 1 - S);
    const bool owner = lane >= 2 && lane < WAVE - 2 && L.x < w;
    L.vst = owner && L.x + 3 < w;
    L.sst = owner && !L.vst;
    L.lo = L.hi = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        L.cx[i] = (unsigned)clampi(L.x + i, S, w - 1 - S);
        if (L.x + i < S) L.lo |= 1u << i;
        if (L.x + i > w - 1 - S) L.hi |= 1u << i;
    }
    return L;
}
template <int N>
__device__ __forceinline__ void det_store(float* const (&plane)[N], size_t ro, const DetLane& L, int w, const f4 (&v)[N]) {
    if (L.vst) {
#pragma unroll
        for (int i = 0; i < N; ++i) plane_store4u(plane[i] + ro + L.x, v[i]);
    } else if (L.sst) {
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (L.x + e < w) plane[i][ro + L.x + e] = v[i][e];
    }
}
template <int S, int N>
__device__ __forceinline__ void det_store_filled(float* const (&plane)[N], const DetLane& L, int w, int h, int y, const f4 (&v)[N]) {
    det_store<N>(plane, (size_t)y * w, L, w, v);
    if (y == S || y == h - 1 - S) {  // wave-uniform, two rows per strip
        if (y == S) {
#pragma nounroll
            for (int r = 0; r < S; ++r) det_store<N>(plane, (size_t)r * w, L, w, v);
        }
        if (y == h - 1 - S) {
#pragma nounroll
            for (int r = h - S; r < h; ++r) det_store<N>(plane, (size_t)r * w, L, w, v);
        }
    }
}
template <int S, bool NMS, bool KEEP>
__global__ void __launch_bounds__(SNT, (S == 1 ? 2 : 1))  // rings + taps need > 256 VGPRs for S >= 2: one wave per SIMD
k_detector_stream(const float* __restrict__ ls, float* __restrict__ lx_out, float* __restrict__ ly_out,
                  float* __restrict__ lxx_out, float* __restrict__ lyy_out, float* __restrict__ lxy_out,
                  float* __restrict__ ldet_out, int w, int h, StreamGrid g, float kn, float kwn, float quat,
                  StreamNms nms) {
    constexpr int P = 2 * S + 1, NOUT = KEEP ? 4 : 1;
    __shared__ f4 s_ring[SNT / WAVE][2][P][WAVE];  // stage-1 H results of the last P rows, per wave
    __shared__ Candidate s_cands[NMS ? SNT / WAVE : 1][NMS ? CAND_BUF : 1];  // per-wave extrema buffer
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    Candidate* const cbuf = s_cands[NMS ? wv : 0];
    unsigned cnum = 0;
    const long wave = (long)blockIdx.x * (SNT / WAVE) + wv;
    if (wave >= g.waves) return;
    // (image, band, strip), strips fastest; bands split the interior rows evenly
    const long per = (long)g.nbands * g.nstrips;
    const int img = (int)(wave / per);
    const int rem = (int)(wave - (long)img * per);
    const int band = rem / g.nstrips, strip = rem - band * g.nstrips;
    const int rows = h - 2 * S;
    const int cs = S + (int)(((long)band * rows) / g.nbands), ce = S + (int)(((long)(band + 1) * rows) / g.nbands);
    if (cs >= ce) return;
    const bool bottom = band == g.nbands - 1;
    const DetLane L = make_det_lane<S>(strip, lane, w);
    const bool edge_strip = strip == 0 || strip == g.nstrips - 1;
    // lane / component that hold columns S and w-1-S in this strip
    const int x0 = strip * (4 * (WAVE - 4)) - 8;
    const int src_lo = clampi((S - x0) >> 2, 0, WAVE - 1), c_lo = (S - x0) & 3;
    const int src_hi = clampi((w - 1 - S - x0) >> 2, 0, WAVE - 1), c_hi = (w - 1 - S - x0) & 3;
    const size_t base = (size_t)img * (size_t)w * (size_t)h;
    const float* in = ls + base;
    float* const out1[2] = {lx_out + base, ly_out + base};
    float* out2[NOUT];
    out2[0] = ldet_out + base;
    if (KEEP) {
        out2[1] = lxx_out + base;
        out2[2] = lyy_out + base;
        out2[3] = lxy_out + base;
    }
    float* const(&out2c)[NOUT] = out2;
    // output rows computed: c_lo_row .. c_hi_row (the extrema test of rows [cs, ce) needs cs-1 and ce too)
    const int c_first = NMS ? max(cs - 1, S) : cs;
    const int c_last = NMS ? min(ce, h - 1 - S) : ce - 1;
    int v0 = c_first - 2 * S;
    int t_last = (bottom ? h - 1 : c_last + 2 * S) - v0;
    if (bottom) {  // the last input row must fall on ring index P-1 (the drain below is compiled only there)
        const int extra = (P - 1 - t_last % P + P) % P;
        v0 -= extra;
        t_last += extra;
    }
    const int T = t_last + 1;
    unsigned xok = 0;  // bit i: pixel i of this lane may be a candidate
    if (NMS && (L.vst || L.sst)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (L.x + i >= nms.xlo && L.x + i <= nms.xhi) xok |= 1u << i;
    }
    auto fetch = [&](int t, f4& fa, f4& fb, f4& fc) {
        const float* row = in + (size_t)clampi(v0 + min(t, T - 1), S, h - 1 - S) * w;
        if (!L.edge) fetch_vec_x<S>(row, L.x, fa, fb, fc);
        else fetch_edge_x<S>(row, L.cx, fa, fb, fc);
    };
    auto fill_cols = [&](f4& q) {  // wave-uniform call sites only (readlane)
        const float lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_lo == 0 ? q[0] : c_lo == 1 ? q[1] : c_lo == 2 ? q[2] : q[3]), src_lo));
        const float hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c_hi == 0 ? q[0] : c_hi == 1 ? q[1] : c_hi == 2 ? q[2] : q[3]), src_hi));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (L.lo & (1u << i)) q[i] = lo;
            if (L.hi & (1u << i)) q[i] = hi;
        }
    };
    f4 rA[P], rB[P], rC[P];
    f4 dm2 = 0.0f, dm1 = 0.0f;
    f4 a, b, c;
    fetch(0, a, b, c);
    for (int t0 = 0; t0 < T; t0 += P) {
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int t = t0 + k;
            if (t < T) {
                const int v = v0 + t;
                f4 na, nb, nc;  // the next row's taps are in flight during this row's arithmetic
                fetch(t + 1, na, nb, nc);
                const f4 hm = tap_main(a, b, c, kn, kwn), ho = tap_off(a, c);
                s_ring[wv][0][k][lane] = hm;
                s_ring[wv][1][k][lane] = ho;
                if (t >= 2 * S) {
                    const int k0 = (k + 1) % P, k1 = (k + P - S) % P;  // ring rows u-S, u; k = row u+S
                    const int u = v - S;
                    const f4 lx = tap_off(s_ring[wv][0][k0][lane], hm);
                    const f4 ly = tap_main(s_ring[wv][1][k0][lane], s_ring[wv][1][k1][lane], ho, kn, kwn);
                    if (u >= cs && u < ce) {
                        const f4 o1[2] = {lx, ly};
                        det_store_filled<S, 2>(out1, L, w, h, u, o1);
                    }
                    f4 xa, xc, ya, yc;
                    shifted<S>(lx, xa, xc);
                    shifted<S>(ly, ya, yc);
                    f4 A = tap_main(xa, lx, xc, kn, kwn);  // H_main(Lx)
                    f4 B = tap_off(ya, yc);                // H_off(Ly)
                    f4 C = tap_off(xa, xc);                // H_off(Lx)
                    if (edge_strip) {
                        fill_cols(A);
                        fill_cols(B);
                        fill_cols(C);
                    }
                    rA[k] = A; rB[k] = B; rC[k] = C;
                    if (u == S) {  // filled Lx / Ly rows 0..S-1 are row S
#pragma unroll
                        for (int d = 1; d <= S; ++d) {
                            rA[(k + P - d) % P] = A; rB[(k + P - d) % P] = B; rC[(k + P - d) % P] = C;
                        }
                    }
                    if (t >= 4 * S) {
                        const int cr = u - S;
                        f4 o[NOUT];
                        const f4 lxx = tap_off(rA[k0], rA[k]);
                        const f4 lyy = tap_main(rB[k0], rB[k1], rB[k], kn, kwn);
                        const f4 lxy = tap_main(rC[k0], rC[k1], rC[k], kn, kwn);
                        const f4 det = ((lxx * lyy) - (lxy * lxy)) * quat;
                        o[0] = det;
                        if (KEEP) { o[1] = lxx; o[2] = lyy; o[3] = lxy; }
                        const f4(&oc)[NOUT] = o;
                        if (cr >= cs && cr < ce) det_store_filled<S, NOUT>(out2c, L, w, h, cr, oc);
                        if (NMS) {
                            const int y = cr - 1;  // row under test: dm1, between dm2 (above) and det (below)
                            if (y >= cs && y < ce && y >= nms.ylo && y <= nms.yhi) {  // uniform
                                const float left = from_left_lane(dm1[3]), right = from_right_lane(dm1[0]);
                                const float xm[4] = {left, dm1[0], dm1[1], dm1[2]};
                                const float xp[4] = {dm1[1], dm1[2], dm1[3], right};
                                unsigned m = 0;
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const float q = dm1[i];
                                    const bool hit = (q > nms.thr) & (q > xp[i]) & (q > xm[i]) & (q > dm2[i]) & (q > det[i]);
                                    m |= hit ? 1u << i : 0u;
                                }
                                m &= xok;
                                wave_cands_push(cbuf, cnum, m, nms, lane, [&](int i) {  // at most two pixels of a quad can be strict maxima
                                    Candidate cd;
                                    cd.level = nms.level;
                                    cd.idx = (unsigned)(y * w + L.x + i);
                                    cd.v = i == 0 ? dm1[0] : i == 1 ? dm1[1] : i == 2 ? dm1[2] : dm1[3];
                                    cd.xp = i == 0 ? dm1[1] : i == 1 ? dm1[2] : i == 2 ? dm1[3] : right;
                                    cd.xm = i == 0 ? left : i == 1 ? dm1[0] : i == 2 ? dm1[1] : dm1[2];
                                    cd.yp = i == 0 ? det[0] : i == 1 ? det[1] : i == 2 ? det[2] : det[3];
                                    cd.ym = i == 0 ? dm2[0] : i == 1 ? dm2[1] : i == 2 ? dm2[2] : dm2[3];
                                    cd.img = (unsigned)img;
                                    return cd;
                                });
                            }
                            dm2 = dm1;
                            dm1 = det;
                        }
                    }
                    if (k == P - 1 && bottom && u == h - 1 - S) {
                        // The filled Lx / Ly rows h-S..h-1 are row h-1-S as well: enter its H result once more per
                        // drained output row (the slot it takes holds the oldest row, which is no longer needed).
#pragma unroll
                        for (int d = 1; d <= S; ++d) {
                            const int cr = u - S + d;
                            const int j2 = (k + d) % P, j1 = (k + d + P - S) % P, j0 = (k + d + 1) % P;
                            rA[j2] = A; rB[j2] = B; rC[j2] = C;
                            f4 o[NOUT];
                            const f4 lxx = tap_off(rA[j0], rA[j2]);
                            const f4 lyy = tap_main(rB[j0], rB[j1], rB[j2], kn, kwn);
                            const f4 lxy = tap_main(rC[j0], rC[j1], rC[j2], kn, kwn);
                            o[0] = ((lxx * lyy) - (lxy * lxy)) * quat;
                            if (KEEP) { o[1] = lxx; o[2] = lyy; o[3] = lxy; }
                            const f4(&oc)[NOUT] = o;
                            if (cr >= cs && cr < ce) det_store_filled<S, NOUT>(out2c, L, w, h, cr, oc);
                        }
                    }
                }
                a = na; b = nb; c = nc;
            }
        }
    }
    if (NMS) wave_cands_flush(cbuf, cnum, nms, lane);
}

// ---------------------------------------------------------------------------------------------
// gaussian_blur with a dense 5-tap kernel (types/image.rs:374-380: V(H(in)), fill_border after each pass), the
// level-0 blur of the pyramid (sigma 1.6).  T = uint8_t folds in create_unit_float_image (types/image.rs:136):
// the 256 possible values of `f32::from(v) * 1f32 / 255f32` are tabulated once per workgroup with that very
// expression, so each tap costs an LDS read instead of a correctly rounded f32 division.
// ---------------------------------------------------------------------------------------------
struct Taps5 {
    float k[5];
};
__device__ __forceinline__ f4 tap5(const f4 (&p)[5], const Taps5& t) {
    f4 acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 5; ++i) acc = acc + t.k[i] * p[i];
    return acc;
}
// the five taps (columns cx-2 .. cx+2) of the lane's four pixels of one input row
template <typename T>
__device__ __forceinline__ void blur5_fetch(const T* __restrict__ row, const Lane& L, const float* __restrict__ lut, f4 (&p)[5]) {
    if (!L.edge) {
        float v[8];  // columns x-2 .. x+5
        if constexpr (sizeof(T) == 1) {
            // x is a multiple of 4 and the caller guarantees 4-byte aligned rows: bytes x-4 .. x+7 are three dwords
            const uint32_t* q = reinterpret_cast<const uint32_t*>(row + L.x - 4);
            const uint32_t a = q[0], b = q[1], c = q[2];
            v[0] = lut[(a >> 16) & 255u]; v[1] = lut[a >> 24];
            v[2] = lut[b & 255u]; v[3] = lut[(b >> 8) & 255u]; v[4] = lut[(b >> 16) & 255u]; v[5] = lut[b >> 24];
            v[6] = lut[c & 255u]; v[7] = lut[(c >> 8) & 255u];
        } else {
            const f4 q0 = *reinterpret_cast<const f4u*>(row + L.x - 2), q1 = *reinterpret_cast<const f4u*>(row + L.x + 2);
            v[0] = q0[0]; v[1] = q0[1]; v[2] = q0[2]; v[3] = q0[3];
            v[4] = q1[0]; v[5] = q1[1]; v[6] = q1[2]; v[7] = q1[3];
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) p[t] = f4{v[t], v[t + 1], v[t + 2], v[t + 3]};
    } else {
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (sizeof(T) == 1) p[t][i] = lut[row[L.cx[i] + t - 2]];
                else p[t][i] = row[L.cx[i] + t - 2];
            }
    }
}

template <typename T>
__global__ void __launch_bounds__(SNT, 4)
k_blur5_stream(const T* __restrict__ in, float* __restrict__ out, int w, int h, StreamGrid g, Taps5 tp) {
    constexpr int S = 2, P = 5;
    __shared__ float s_lut[256];
    if (sizeof(T) == 1) {
        s_lut[threadIdx.x & 255] = ((float)(threadIdx.x & 255) * 1.0f) / 255.0f;  // SNT == 256
        __syncthreads();
    }
    const int lane = threadIdx.x & (WAVE - 1);
    const long wave = wave_index();
    if (wave >= g.waves) return;
    const Piece pc = wave_piece(wave, g, S, h);
    if (pc.cs >= pc.ce) return;
    const Lane L = make_lane<S, 0>(pc.strip, lane, w);
    const size_t base = (size_t)pc.img * (size_t)w * (size_t)h;
    const T* src = in + base;
    float* const dst[1] = {out + base};
    const int v0 = pc.cs - S, T_ = (pc.ce - pc.cs) + 2 * S;
    f4 ring[P];
    f4 cur[5], nxt[5];
    blur5_fetch<T>(src + (size_t)clampi(v0, S, h - 1 - S) * w, L, s_lut, cur);
    for (int t0 = 0; t0 < T_; t0 += P) {
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int t = t0 + k;
            if (t < T_) {
                blur5_fetch<T>(src + (size_t)clampi(v0 + min(t + 1, T_ - 1), S, h - 1 - S) * w, L, s_lut, nxt);
                ring[k] = tap5(cur, tp);
                if (t >= 2 * S) {
                    const f4 col[5] = {ring[(k + 1) % P], ring[(k + 2) % P], ring[(k + 3) % P], ring[(k + 4) % P], ring[k]};
                    const f4 v[1] = {tap5(col, tp)};
                    store_filled<S, 1>(dst, L, w, h, v0 + t - S, v);
                }
#pragma unroll
                for (int i = 0; i < 5; ++i) cur[i] = nxt[i];
            }
        }
    }
}

// Bands are sized so that one wave per resident slot covers the batch in a single round.
template <typename K>
inline StreamGrid plan_stream(K kernel, uint32_t w, uint32_t h, uint32_t n, int S, int HL, int min_rows, dim3* grid) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    static int per_cu = 0;  // one per kernel instantiation
    if (!per_cu && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, SNT, 0) != hipSuccess || per_cu <= 0))
        per_cu = 2;
    const int sw = 4 * (WAVE - 2 * HL);
    const long slots = (long)cus * per_cu * (SNT / WAVE);
    const int rows = (int)h - 2 * S;
    StreamGrid g;
    g.nstrips = (int)((w + sw - 1) / sw);
    const long cols = (long)n * g.nstrips;
    // enough bands to give every resident slot a wave; a band restarts the ring (warm-up rows), so bands stay
    // at least min_rows tall even when that leaves slots idle (small levels are latency-bound either way)
    const long nb_fill = std::max<long>(1, (slots + cols - 1) / cols);
    g.band_rows = (int)std::max<long>((rows + nb_fill - 1) / nb_fill, std::min<long>(min_rows, rows));
    g.nbands = (rows + g.band_rows - 1) / g.band_rows;
    g.waves = cols * g.nbands;
    *grid = dim3((unsigned)((g.waves + SNT / WAVE - 1) / (SNT / WAVE)));
    return g;
}

}  // namespace

namespace launch {

// The candidate rectangle of scale_space_extrema.rs:32-42 (x in 1..w-2, y in 1..h-2) and :80-87 (descriptor
// window inside the image) with the reference's float expressions; both border tests are monotone in the
// coordinate, so the admissible coordinates form one interval.
static void admissible(uint32_t dim, float border_m, int* lo, int* hi) {
    auto out_lo = [&](int v) { return (roundf((float)v - border_m) - 1.0f) < 0.0f; };
    auto out_hi = [&](int v) { return (roundf((float)v + border_m) + 1.0f) >= (float)dim; };
    int a = 1, b = (int)dim - 2;
    while (a <= b && out_lo(a)) ++a;
    while (b >= a && out_hi(b)) --b;
    *lo = a;
    *hi = b;  // empty when lo > hi
}

bool detector_stream_supported(uint32_t sigma, uint32_t w, uint32_t h, float border_m, bool nms) {
    if (sigma < 1 || sigma > 4) return false;
    if (w < 4 * sigma + 8 || h < 4 * sigma + 8) return false;
    // the extrema test reads Ldet one pixel around a candidate: keep that ring inside the interior rows/columns
    return !nms || border_m >= (float)(sigma + 2);
}

// 5-tap gaussian_blur as a streaming kernel; u8 input needs 4-byte aligned rows (w % 4 == 0) for its dword loads
bool blur5_stream_supported(uint32_t w, uint32_t h, uint32_t ntaps, bool is_u8) {
    return ntaps == 5 && w >= 16 && h >= 16 && (!is_u8 || (w & 3u) == 0);
}
template <typename T>
static void blur5_stream_t(hipStream_t s, const T* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k) {
    Taps5 tp;
    for (int i = 0; i < 5; ++i) tp.k[i] = k[i];
    dim3 grid;
    const StreamGrid g = plan_stream(k_blur5_stream<T>, w, h, n, 2, 0, 12, &grid);
    hipLaunchKernelGGL((k_blur5_stream<T>), grid, dim3(SNT), 0, s, in, out, (int)w, (int)h, g, tp);
}
void blur5_stream_u8(hipStream_t s, const uint8_t* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k) {
    blur5_stream_t<uint8_t>(s, in, out, w, h, n, k);
}
void blur5_stream_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k) {
    blur5_stream_t<float>(s, in, out, w, h, n, k);
}

bool prep_stream_supported(uint32_t w, uint32_t h) { return w >= 16 && h >= 16; }
static int prep_min_rows() {
    static int v = 0;
    if (!v) {
        const char* e = getenv("AKZ_PREP_MIN_ROWS");  // tuning knob
        v = e ? std::max(1, atoi(e)) : 8;
    }
    return v;
}

void prep_stream(hipStream_t s, const float* prev, bool half, float* lt_out, float* lsmooth, float* lflow, uint32_t w,
                 uint32_t h, uint32_t pw, uint32_t ph, uint32_t n, const float* g3, const double* d_k, uint32_t k_pow) {
    const Taps m = taps_scharr_main(1);
    dim3 grid;
    if (half) {
        const StreamGrid g = plan_stream(k_prep_stream<true, 0>, w, h, n, 1, 1, prep_min_rows(), &grid);
        hipLaunchKernelGGL((k_prep_stream<true, 0>), grid, dim3(SNT), 0, s, prev, lt_out, lsmooth, lflow, (int)w, (int)h,
                           (int)pw, (int)ph, g, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow, ContrastArgs{});
    } else {
        const StreamGrid g = plan_stream(k_prep_stream<false, 0>, w, h, n, 1, 1, prep_min_rows(), &grid);
        hipLaunchKernelGGL((k_prep_stream<false, 0>), grid, dim3(SNT), 0, s, prev, lt_out, lsmooth, lflow, (int)w, (int)h,
                           (int)pw, (int)ph, g, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow, ContrastArgs{});
    }
}

// compute_contrast_factor's two passes over gaussian_blur(in, sigma with the 3 taps g3) without materialising the
// blurred plane: maximum, then histogram (d_hmax_bits / d_hist zeroed by the caller).  nbins <= 1024.
bool contrast_stream_supported(uint32_t w, uint32_t h, uint32_t ntaps, uint32_t nbins) {
    return ntaps == 3 && nbins <= 1024 && prep_stream_supported(w, h);
}
void contrast_stream(hipStream_t s, const float* in, uint32_t w, uint32_t h, uint32_t n, const float* g3,
                     unsigned long long* d_hmax_bits, uint32_t nbins, uint32_t* d_hist) {
    const Taps m = taps_scharr_main(1);
    const ContrastArgs ca{d_hmax_bits, d_hist, nbins};
    dim3 grid;
    const StreamGrid g1 = plan_stream(k_prep_stream<false, 1>, w, h, n, 1, 1, prep_min_rows(), &grid);
    hipLaunchKernelGGL((k_prep_stream<false, 1>), grid, dim3(SNT), 0, s, in, nullptr, nullptr, nullptr, (int)w, (int)h,
                       (int)w, (int)h, g1, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], nullptr, 0u, ca);
    const StreamGrid g2 = plan_stream(k_prep_stream<false, 2>, w, h, n, 1, 1, prep_min_rows(), &grid);
    hipLaunchKernelGGL((k_prep_stream<false, 2>), grid, dim3(SNT), (SNT / WAVE) * CHIST_COPIES * nbins * sizeof(unsigned), s,
                       in, nullptr, nullptr, nullptr, (int)w, (int)h, (int)w, (int)h, g2, g3[0], g3[1], g3[2], m.wgt[0],
                       m.wgt[1], nullptr, 0u, ca);
}

#define AKZ_FDET(S)                                                                                                   \
    case S: {                                                                                                         \
        dim3 gr;                                                                                                      \
        if (d_cand && keep) {                                                                                         \
            const StreamGrid sg = plan_stream(k_detector_stream<S, true, true>, w, h, n, S, 2, 4 * (S + 1), &gr);      \
            hipLaunchKernelGGL((k_detector_stream<S, true, true>), gr, dim3(SNT), 0, s, lsmooth, lx, ly, lxx, lyy, lxy, \
                               ldet_out, (int)w, (int)h, sg, kn, kwn, quat, na);                                      \
        } else if (d_cand) {                                                                                          \
            const StreamGrid sg = plan_stream(k_detector_stream<S, true, false>, w, h, n, S, 2, 4 * (S + 1), &gr);     \
            hipLaunchKernelGGL((k_detector_stream<S, true, false>), gr, dim3(SNT), 0, s, lsmooth, lx, ly, lxx, lyy,     \
                               lxy, ldet_out, (int)w, (int)h, sg, kn, kwn, quat, na);                                 \
        } else if (keep) {                                                                                            \
            const StreamGrid sg = plan_stream(k_detector_stream<S, false, true>, w, h, n, S, 2, 4 * (S + 1), &gr);     \
            hipLaunchKernelGGL((k_detector_stream<S, false, true>), gr, dim3(SNT), 0, s, lsmooth, lx, ly, lxx, lyy,     \
                               lxy, ldet_out, (int)w, (int)h, sg, kn, kwn, quat, na);                                 \
        } else {                                                                                                      \
            const StreamGrid sg = plan_stream(k_detector_stream<S, false, false>, w, h, n, S, 2, 4 * (S + 1), &gr);    \
            hipLaunchKernelGGL((k_detector_stream<S, false, false>), gr, dim3(SNT), 0, s, lsmooth, lx, ly, lxx, lyy,    \
                               lxy, ldet_out, (int)w, (int)h, sg, kn, kwn, quat, na);                                 \
        }                                                                                                             \
    } break;

// Fused streaming detector of one level (k_detector_stream): same contract as detector_stream.
void detector_fused_stream(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx,
                           float* lyy, float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n, uint32_t level,
                           float thr, float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count) {
    const Taps m = taps_scharr_main(sigma);
    const float kn = m.wgt[0], kwn = m.wgt[1];
    const float quat = (float)(sigma * sigma * sigma * sigma);
    const bool keep = lxx && lyy && lxy;
    StreamNms na{level, thr, 0, -1, 0, -1, d_cand, cap, d_count};
    if (d_cand) {
        admissible(w, border_m, &na.xlo, &na.xhi);
        admissible(h, border_m, &na.ylo, &na.yhi);
    }
    switch (sigma) {
        AKZ_FDET(1) AKZ_FDET(2) AKZ_FDET(3) AKZ_FDET(4)
        default: break;
    }
}
#undef AKZ_FDET

#define AKZ_SDET(S)                                                                                                  \
    case S: {                                                                                                        \
        dim3 g1, g2;                                                                                                 \
        const StreamGrid s1 = plan_stream(k_deriv1_stream<S>, w, h, n, S, 0, 4 * (S + 1), &g1);                                   \
        hipLaunchKernelGGL((k_deriv1_stream<S>), g1, dim3(SNT), 0, s, lsmooth, lx, ly, (int)w, (int)h, s1, kn, kwn); \
        if (d_cand) {                                                                                                \
            if (keep) {                                                                                              \
                const StreamGrid s2 = plan_stream(k_deriv2_stream<S, true, true>, w, h, n, S, 1, 4 * (S + 1), &g2);               \
                hipLaunchKernelGGL((k_deriv2_stream<S, true, true>), g2, dim3(SNT), 0, s, (const float*)lx,          \
                                   (const float*)ly, lxx, lyy, lxy, ldet_out, (int)w, (int)h, s2, kn, kwn, quat, na); \
            } else {                                                                                                 \
                const StreamGrid s2 = plan_stream(k_deriv2_stream<S, true, false>, w, h, n, S, 1, 4 * (S + 1), &g2);              \
                hipLaunchKernelGGL((k_deriv2_stream<S, true, false>), g2, dim3(SNT), 0, s, (const float*)lx,         \
                                   (const float*)ly, lxx, lyy, lxy, ldet_out, (int)w, (int)h, s2, kn, kwn, quat, na); \
            }                                                                                                        \
        } else if (keep) {                                                                                           \
            const StreamGrid s2 = plan_stream(k_deriv2_stream<S, false, true>, w, h, n, S, 0, 4 * (S + 1), &g2);                  \
            hipLaunchKernelGGL((k_deriv2_stream<S, false, true>), g2, dim3(SNT), 0, s, (const float*)lx,             \
                               (const float*)ly, lxx, lyy, lxy, ldet_out, (int)w, (int)h, s2, kn, kwn, quat, na);    \
        } else {                                                                                                     \
            const StreamGrid s2 = plan_stream(k_deriv2_stream<S, false, false>, w, h, n, S, 0, 4 * (S + 1), &g2);                 \
            hipLaunchKernelGGL((k_deriv2_stream<S, false, false>), g2, dim3(SNT), 0, s, (const float*)lx,            \
                               (const float*)ly, lxx, lyy, lxy, ldet_out, (int)w, (int)h, s2, kn, kwn, quat, na);    \
        }                                                                                                            \
    } break;

// Streaming detector of one level: first derivatives, second derivatives + Ldet and, when d_cand is
// given, the extrema candidates.  lxx/lyy/lxy may be null together (the planes are then not kept).
void detector_stream(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx, float* lyy,
                     float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n, uint32_t level, float thr,
                     float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count) {
    const Taps m = taps_scharr_main(sigma);
    const float kn = m.wgt[0], kwn = m.wgt[1];
    const float quat = (float)(sigma * sigma * sigma * sigma);
    const bool keep = lxx && lyy && lxy;
    StreamNms na{level, thr, 0, -1, 0, -1, d_cand, cap, d_count};
    if (d_cand) {
        admissible(w, border_m, &na.xlo, &na.xhi);
        admissible(h, border_m, &na.ylo, &na.yhi);
    }
    switch (sigma) {
        AKZ_SDET(1) AKZ_SDET(2) AKZ_SDET(3) AKZ_SDET(4)
        default: break;
    }
}
#undef AKZ_SDET

}  // namespace launch
}  // namespace akz
