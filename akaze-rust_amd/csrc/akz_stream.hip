// Barrier-free streaming kernels for gfx950: the level preparation (lib.rs:80-105), the contrast-factor passes
// (contrast_factor.rs:18-71) and the level-0 blur (lib.rs:56) as register-ring stencil chains.
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
// preparation kernel is ~2x faster than the LDS-tiled one for batches (4.2 vs 2.2 TB/s at 32 x 1080p).  The
// detector's streaming forms of round 1 (a kernel pair and a fused kernel with four pixels per lane, which needed
// more than 256 VGPRs) were replaced by the workgroup-wide column march of akz_march.hip.
//
// Arithmetic is the reference's: f32 mul then add, taps left to right starting from 0.0f, no FMA.
// The off-axis Scharr taps [-1, 0.., 0, ..0, 1] are evaluated as (0.0f - a) + c, which is
// bit-identical to ((0.0f + -1.0f*a) + 0.0f*b) + 1.0f*c for every finite b (the zero taps add +-0
// to an accumulator that is never -0).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "akz_internal.hpp"
#include "akz_pm_g2.hpp"

namespace akz {
namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // four pixels of a row, dword-aligned

constexpr int WAVE = 64, SNT = 256;  // 4 independent waves per workgroup

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
    const double* thr;              // MODE 2: per image, nbins + 1 bin thresholds on Lx^2 + Ly^2 (k_contrast_thresholds)
};
constexpr int CHIST_COPIES = 4;  // sub-histograms per wave (lanes spread over them) against same-bin conflicts

// The histogram bin of a pixel with ss = Lx^2 + Ly^2 (f64), exactly as contrast_factor.rs:49-57 computes it.
__device__ __forceinline__ unsigned contrast_bin(double ss, double hmax, unsigned nbins) {
    const double gm = sqrt(ss);
    const double f = floor((double)nbins * (gm / hmax));
    return f >= (double)nbins ? nbins - 1u : (f > 0.0 ? (unsigned)f : 0u);
}
// thr[img][b] = the smallest ss >= 0 (+inf included) whose bin is >= b; NaN if there is none, and for b = nbins.
// contrast_bin is non-decreasing in ss (correctly rounded sqrt, division by a constant, multiplication, floor), and
// non-negative doubles order like their bit patterns: a 64-step bisection per threshold.  One workgroup per image.
__global__ void k_contrast_thresholds(const unsigned long long* __restrict__ hmax_bits, unsigned nbins, double* __restrict__ thr) {
    const double hmax = __longlong_as_double((long long)hmax_bits[blockIdx.x]);
    double* out = thr + (size_t)blockIdx.x * (nbins + 1);
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (unsigned b = threadIdx.x; b <= nbins; b += blockDim.x) {
        if (b == 0) { out[0] = 0.0; continue; }
        if (b == nbins) { out[b] = nan; continue; }
        unsigned long long lo = 0ull, hi = 0x7ff0000000000000ull;  // bit patterns of +0.0 and +inf
        if (!(contrast_bin(__longlong_as_double((long long)hi), hmax, nbins) >= b)) { out[b] = nan; continue; }
        {   // the threshold lies next to (b hmax / nbins)^2: a bracket of +-2^-30 around it, if it is one, saves half the steps
            const double x = (double)b * hmax / (double)nbins, c = x * x;
            const double a = c * (1.0 - 0x1p-30), d = c * (1.0 + 0x1p-30);
            if (a > 0.0 && d < __longlong_as_double(0x7fe0000000000000ll) && contrast_bin(a, hmax, nbins) < b &&
                contrast_bin(d, hmax, nbins) >= b) {
                lo = (unsigned long long)__double_as_longlong(a) + 1ull;
                hi = (unsigned long long)__double_as_longlong(d);
            }
        }
        while (lo < hi) {  // invariant: pred(hi) holds
            const unsigned long long mid = lo + ((hi - lo) >> 1);
            if (contrast_bin(__longlong_as_double((long long)mid), hmax, nbins) >= b) hi = mid;
            else lo = mid + 1;
        }
        out[b] = __longlong_as_double((long long)hi);
    }
}

template <bool HALF, int MODE>
__global__ void __launch_bounds__(SNT, 3)
k_prep_stream(const float* __restrict__ prev, float* __restrict__ lt_out, float* __restrict__ lsmooth,
              float* __restrict__ lflow, int w, int h, int pw, int ph, StreamGrid g, float g0, float g1, float g2,
              float kn, float kwn, const double* __restrict__ d_k, unsigned k_pow, ContrastArgs ca) {
    static_assert(!(HALF && MODE != 0), "the contrast passes read the level itself");
    extern __shared__ __attribute__((aligned(16))) unsigned s_chist[];  // MODE 2: [wave][copy][bin], then [wave][bin thresholds]
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
    double inverse_k = 0.0, hmax = 0.0, gmax = 0.0, bin_scale = 0.0;
    unsigned* myhist = nullptr;
    const double* mythr = nullptr;
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
            // the image's bin thresholds, wave-private too (behind the histograms of all waves: an even number of words, so 8-byte aligned for every bin count)
            double* wt = reinterpret_cast<double*>(s_chist + (size_t)(SNT / WAVE) * CHIST_COPIES * ca.nbins) +
                         (size_t)(threadIdx.x >> 6) * (ca.nbins + 1);
            for (unsigned b = lane; b <= ca.nbins; b += WAVE) wt[b] = ca.thr[(size_t)pc.img * (ca.nbins + 1) + b];
            mythr = wt;
            bin_scale = (double)ca.nbins * (1.0 / hmax);
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
                                // bin = contrast_bin(ss, hmax, nbins), a non-decreasing function of ss: a guess from
                                // the hardware's approximate square root (off by one bin at most), settled by the
                                // exact thresholds of the two neighbouring bins -- instead of a correctly rounded f64
                                // square root and division per pixel
                                if (ss != 0.0) {  // sqrt(ss) != 0.0 (contrast_factor.rs:51)
                                    const double ga = __builtin_amdgcn_sqrt(ss) * bin_scale;
                                    const unsigned g = min(ga > 0.0 ? (unsigned)ga : 0u, ca.nbins - 1u);  // NaN -> 0
                                    unsigned b = g;
                                    if (ss < mythr[g]) b = g - 1u;
                                    else if (ss >= mythr[g + 1u]) b = g + 1u;
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
static int prep_min_rows() { return 8; }

void prep_stream(hipStream_t s, const float* prev, bool half, float* lt_out, float* lsmooth, float* lflow, uint32_t w,
                 uint32_t h, uint32_t pw, uint32_t ph, uint32_t n, const float* g3, const double* d_k, uint32_t k_pow) {
    const Taps m = taps_scharr_main(1);
    dim3 grid;
    if (half) {
        const StreamGrid g = plan_stream(k_prep_stream<true, 0>, w, h, n, 1, 1, prep_min_rows(), &grid);
        hipLaunchKernelGGL((k_prep_stream<true, 0>), grid, dim3(SNT), 0, s, prev, lt_out, lsmooth, lflow, (int)w, (int)h,
                           (int)pw, (int)ph, g, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow, ContrastArgs{nullptr, nullptr, 0u, nullptr});
    } else {
        const StreamGrid g = plan_stream(k_prep_stream<false, 0>, w, h, n, 1, 1, prep_min_rows(), &grid);
        hipLaunchKernelGGL((k_prep_stream<false, 0>), grid, dim3(SNT), 0, s, prev, lt_out, lsmooth, lflow, (int)w, (int)h,
                           (int)pw, (int)ph, g, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow, ContrastArgs{nullptr, nullptr, 0u, nullptr});
    }
}

// compute_contrast_factor's two passes over gaussian_blur(in, sigma with the 3 taps g3) without materialising the
// blurred plane: maximum, bin thresholds, histogram (d_hmax_bits / d_hist zeroed by the caller; d_thr: n x (nbins + 1)
// doubles).  nbins <= 640.
bool contrast_stream_supported(uint32_t w, uint32_t h, uint32_t ntaps, uint32_t nbins) {
    return ntaps == 3 && nbins <= 640 && prep_stream_supported(w, h);  // 61 KB of LDS at 640 bins
}
void contrast_thresholds(hipStream_t s, const unsigned long long* d_hmax_bits, uint32_t nbins, uint32_t n, double* d_thr) {
    hipLaunchKernelGGL(k_contrast_thresholds, dim3(n), dim3(320), 0, s, d_hmax_bits, nbins, d_thr);
}
void contrast_stream(hipStream_t s, const float* in, uint32_t w, uint32_t h, uint32_t n, const float* g3,
                     unsigned long long* d_hmax_bits, uint32_t nbins, uint32_t* d_hist, double* d_thr) {
    const Taps m = taps_scharr_main(1);
    const ContrastArgs ca{d_hmax_bits, d_hist, nbins, d_thr};
    dim3 grid;
    const StreamGrid g1 = plan_stream(k_prep_stream<false, 1>, w, h, n, 1, 1, prep_min_rows(), &grid);
    hipLaunchKernelGGL((k_prep_stream<false, 1>), grid, dim3(SNT), 0, s, in, nullptr, nullptr, nullptr, (int)w, (int)h,
                       (int)w, (int)h, g1, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], nullptr, 0u, ca);
    contrast_thresholds(s, d_hmax_bits, nbins, n, d_thr);
    const StreamGrid g2 = plan_stream(k_prep_stream<false, 2>, w, h, n, 1, 1, prep_min_rows(), &grid);
    const size_t lds = ((size_t)(SNT / WAVE) * CHIST_COPIES * nbins) * sizeof(unsigned) +
                       (size_t)(SNT / WAVE) * (nbins + 1) * sizeof(double);
    hipLaunchKernelGGL((k_prep_stream<false, 2>), grid, dim3(SNT), lds, s,
                       in, nullptr, nullptr, nullptr, (int)w, (int)h, (int)w, (int)h, g2, g3[0], g3[1], g3[2], m.wgt[0],
                       m.wgt[1], nullptr, 0u, ca);
}

}  // namespace launch
}  // namespace akz
