// gfx950 kernels of the A-KAZE hot path.  All image arithmetic is IEEE f32/f64 add/mul/div/sqrt
// in the reference's evaluation order; the file must be compiled with -ffp-contract=off and
// correctly rounded f32 divide/sqrt, f32 denormals on (see Makefile).  No MFMA: every kernel here
// is a <2 flop/byte stencil, gather or popcount loop.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "akz_internal.hpp"
#include "akz_pm_g2.hpp"
#include "akz_libm.hpp"
#include "akz_prep_passes.hpp"

namespace akz {
namespace {

constexpr int BX = 64, BY = 4;  // one wave per image row segment, 4 rows per workgroup

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ float px(const float* p, long i) { return p[i]; }
// create_unit_float_image: f32::from(v) * 1f32 / 255f32  (akaze/src/types/image.rs:136)
__device__ __forceinline__ float px(const uint8_t* p, long i) { return ((float)p[i] * 1.0f) / 255.0f; }

// ---------------------------------------------------------------------------------------------
// horizontal_filter / vertical_filter + fill_border  (akaze/src/types/image.rs:239-332)
// out(x,y) = raw(clamp(x,hw,w-1-hw), clamp(y,hw,h-1-hw)); raw = ((0 + k0*I0) + k1*I1) + ...
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void k_filter_h(const T* __restrict__ in, float* __restrict__ out, int w, int h, Taps t) {
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    const int cx = clampi(x, t.hw, w - 1 - t.hw), cy = clampi(y, t.hw, h - 1 - t.hw);
    const T* row = in + base + (size_t)cy * w + cx;
    float acc = 0.0f;
    for (int i = 0; i < t.n; ++i) acc = acc + t.wgt[i] * px(row, t.off[i]);
    out[base + (size_t)y * w + x] = acc;
}

__global__ void k_filter_v(const float* __restrict__ in, float* __restrict__ out, int w, int h, Taps t) {
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    const int cx = clampi(x, t.hw, w - 1 - t.hw), cy = clampi(y, t.hw, h - 1 - t.hw);
    const float* col = in + base + (size_t)cy * w + cx;
    float acc = 0.0f;
    for (int i = 0; i < t.n; ++i) acc = acc + t.wgt[i] * col[(long)t.off[i] * w];
    out[base + (size_t)y * w + x] = acc;
}

// One wave that spins for `ticks` of the constant 100 MHz counter and exits (stream-placement probe of akz_api.cpp: two
// streams whose delays add up share a hardware queue).  Every lane reaches the exit: the loop is bounded by the clock.
__global__ void k_delay(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned spins = 0;
    while (wall_clock64() - t0 < ticks && spins < (1u << 26)) ++spins;
    if (sink && spins == 0xffffffffu) *sink = spins;  // (keeps the loop)
}
// half_size for widths that are multiples of 8 (source rows and output pairs 16- / 8-byte aligned): a thread reads two
// float4 (source rows 2y and 2y+1, four columns) and writes two outputs -- the one-pixel form below reads the source as
// 8-byte pairs with a stride of two, which reached 1.45 TB/s on a 32-frame half-resolution level (57 us).
__global__ void k_half_size4(const float* __restrict__ in, float* __restrict__ out, int w, int h) {
    const int ow = w / 2, oh = h / 2;
    const int xp = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;  // xp: pair of output columns
    if (2 * xp >= ow || y >= oh) return;
    const float* src = in + (size_t)blockIdx.z * (size_t)w * (size_t)h + (size_t)(2 * y) * w + 4 * xp;
    const float4 t = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + w);
    float v0 = 0.0f, v1 = 0.0f;
    v0 = v0 + t.x; v0 = v0 + b.x; v0 = v0 + t.y; v0 = v0 + b.y;
    v1 = v1 + t.z; v1 = v1 + b.z; v1 = v1 + t.w; v1 = v1 + b.w;
    *reinterpret_cast<float2*>(out + (size_t)blockIdx.z * (size_t)ow * (size_t)oh + (size_t)y * ow + 2 * xp) = make_float2(v0 / 4.0f, v1 / 4.0f);
}
// half_size (akaze/src/types/image.rs:102-118): (((0+a00)+a01)+a10)+a11)/4, a01 = (2x, 2y+1)
__global__ void k_half_size(const float* __restrict__ in, float* __restrict__ out, int w, int h) {
    const int ow = w / 2, oh = h / 2;
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= ow || y >= oh) return;
    const float* src = in + (size_t)blockIdx.z * (size_t)w * (size_t)h;
    float v = 0.0f;
    v = v + src[(size_t)(2 * y) * w + 2 * x];
    v = v + src[(size_t)(2 * y + 1) * w + 2 * x];
    v = v + src[(size_t)(2 * y) * w + 2 * x + 1];
    v = v + src[(size_t)(2 * y + 1) * w + 2 * x + 1];
    out[(size_t)blockIdx.z * (size_t)ow * (size_t)oh + (size_t)y * ow + x] = v / 4.0f;
}

// pm_g2 (akaze/src/lib.rs:26-41), f64 inside; the contrast factor of octave o is k*0.75*...*0.75
// multiplied one octave at a time in f64 as lib.rs:84 does.
__device__ __forceinline__ double octave_contrast(double k, unsigned pow) {
    for (unsigned i = 0; i < pow; ++i) k = k * 0.75;
    return k;
}
// test hook: the reciprocal of pm_g2 alone, on doubles the test chooses (akz_pm_g2.hpp)
__global__ void k_rcp_f64_to_f32(const double* __restrict__ x, float* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = rcp_f64_to_f32(x[i]);
}
__global__ void k_libm_eval(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out3, size_t n, unsigned fma,
                            unsigned* __restrict__ flag) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool bad = false;
    out3[3 * i] = libm::atan2f_glibc(a[i], b[i]);
    out3[3 * i + 1] = fma ? libm::cosf_glibc<true>(a[i], &bad) : libm::cosf_glibc<false>(a[i], &bad);
    out3[3 * i + 2] = fma ? libm::sinf_glibc<true>(a[i], &bad) : libm::sinf_glibc<false>(a[i], &bad);
    if (bad && flag) atomicOr(flag, 1u);
}
__global__ void k_pm_g2(const float* __restrict__ lx, const float* __restrict__ ly, float* __restrict__ out,
                        size_t plane, const double* __restrict__ d_k, unsigned pow) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane) return;
    const double k = octave_contrast(d_k[blockIdx.z], pow);
    const double inverse_k = 1.0 / (k * k);
    const size_t o = (size_t)blockIdx.z * plane + i;
    out[o] = pm_g2_px(lx[o], ly[o], inverse_k);
}

// scale-1 Scharr pair evaluated from a plane with the reference's two-level border clamp
// (derivatives.rs:41-65 over image.rs:270-332).  "lx" = V_off(H_main(I)), "ly" = V_main(H_off(I)).
struct Scharr1 {
    float n, wn;  // main-axis taps [n, wn, n]
};
__device__ __forceinline__ void scharr1_pair(const float* __restrict__ I, int w, int h, int x, int y, Scharr1 k,
                                             float& lx, float& ly) {
    const int cx = clampi(x, 1, w - 2), cy = clampi(y, 1, h - 2);
    float hm[3], ho[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int ry = clampi(cy + r - 1, 1, h - 2);  // fill_border of the H pass
        const float* p = I + (size_t)ry * w + cx;
        const float a = p[-1], b = p[0], c = p[1];
        hm[r] = ((0.0f + k.n * a) + k.wn * b) + k.n * c;
        ho[r] = ((0.0f + -1.0f * a) + 0.0f * b) + 1.0f * c;
    }
    lx = ((0.0f + -1.0f * hm[0]) + 0.0f * hm[1]) + 1.0f * hm[2];
    ly = ((0.0f + k.n * ho[0]) + k.wn * ho[1]) + k.n * ho[2];
}

// Lsmooth -> Lflow  (akaze/src/lib.rs:98-105)
__global__ void k_flow(const float* __restrict__ ls, float* __restrict__ out, int w, int h, Scharr1 sk,
                       const double* __restrict__ d_k, unsigned pow) {
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    float lx, ly;
    scharr1_pair(ls + base, w, h, x, y, sk, lx, ly);
    const double k = octave_contrast(d_k[blockIdx.z], pow);
    out[base + (size_t)y * w + x] = pm_g2_px(lx, ly, 1.0 / (k * k));
}

// ---------------------------------------------------------------------------------------------
// FED step, direct form (akaze/src/ops/nonlinear_diffusion.rs:15-144).  Jacobi update: reads
// the pre-step Lt, writes Lt' (ping-pong) and optionally Lstep.
// ---------------------------------------------------------------------------------------------
__global__ void k_fed_step(const float* __restrict__ L, const float* __restrict__ C, float* __restrict__ Lout,
                           float* __restrict__ Lstep, int w, int h, float half_tau) {
    const int x = blockIdx.x * BX + threadIdx.x, y = blockIdx.y * BY + threadIdx.y;
    if (x >= w || y >= h) return;
    const size_t i = (size_t)blockIdx.z * (size_t)w * (size_t)h + (size_t)y * w + x;
    const float l = L[i], c = C[i];
    const bool hxp = x + 1 < w, hxn = x > 0, hyp = y + 1 < h, hyn = y > 0;
    float xpos = 0.0f, xneg = 0.0f;
    if (hxp) xpos = (c + C[i + 1]) * (L[i + 1] - l);
    if (hxn) xneg = (C[i - 1] + c) * (l - L[i - 1]);
    float t = hxp ? (hxn ? xpos - xneg : xpos) : -xneg;
    if (hyp) {
        t = t + (c + C[i + w]) * (L[i + w] - l);
        if (hyn) t = t - (C[i - w] + c) * (l - L[i - w]);
    } else {
        t = t + (c + C[i - w]) * (L[i - w] - l);  // last row: y_pos taken towards y-1 (:104-119)
    }
    const float st = half_tau * t;
    if (Lstep) Lstep[i] = st;
    Lout[i] = l + st;
}

// ---------------------------------------------------------------------------------------------
// FED, LDS-tiled and temporally fused: one launch advances a tile by n <= HALO explicit steps.
// The workgroup loads its TW x TH tile plus a halo of n rows (and HALO columns, kept a multiple
// of 4 so that every LDS/global access is a 16-byte float4) of Lt and Lflow into LDS, runs the n
// Jacobi steps between two LDS copies of Lt, and stores only the centre.  After step s the values
// within (n - s) pixels of the centre are exact, further out they are stale and never used, so the
// centre after step n equals n launches of k_fed_step bit for bit.  Lflow is constant across the
// steps of a level (lib.rs:105-118), so HBM traffic per step drops from 12 B/px to
// (8*overfetch + 4)/n B/px.  East/west fluxes are shared between neighbours: x_neg(x) and
// x_pos(x-1) are the same expression (nonlinear_diffusion.rs:63-64), evaluated once.
// ---------------------------------------------------------------------------------------------
struct FedTaus {
    int n;
    float half_tau[16];  // 0.5f * (tau as f32) per step (nonlinear_diffusion.rs:67); 8 used by all but k_fed_own<.,.,16,.>
};

// neighbour lane exchange as a one-instruction DPP move (no LDS crossbar round trip as with ds_bpermute)
__device__ __forceinline__ float from_left_lane(float v) {  // lane i receives lane i-1 (DPP wave_shr:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_right_lane(float v) {  // lane i receives lane i+1 (DPP wave_shl:1)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// ---------------------------------------------------------------------------------------------
// FED with register ownership (the default).  Tile / halo / bit-exactness argument as above; data movement:
// every thread OWNS two vertically adjacent float4 groups
// of the region for all n fused steps.  Its Lt values stay in registers, the Lflow pair sums
// (c + c_neighbour), which are constant over the steps of a level, are formed once, left/right
// neighbours come from the adjacent lanes by a DPP move, and LDS is touched only for the row above and
// the row below (2 x b128 read + 2 x b128 write per thread and step; a first version that moved every value
// through LDS spent 66 % of its LDS cycles in bank conflicts and was removed).  All threads run every step on the whole region (stale
// values further than n - s pixels from the centre are never used), so there is no divergence
// around the shuffles.
// ---------------------------------------------------------------------------------------------
#ifndef AKZ_FED_WAVES
#define AKZ_FED_WAVES 6  // 80 VGPRs: three 512-thread workgroups per CU instead of two (+9 % on the FED launches); 8 would spill
#endif
// EPI: the NEXT level's preparation as an epilogue of this level's last diffusion launch (a lone frame's chain is ~45
// dependent launches of 6-16 us, most of which is what a launch costs at that size, not its work: profiles/r06_lone_libm.txt).
// The next level of the same octave starts from this level's final Lt (a clone, lib.rs:92) and its Lsmooth / Lflow need that
// plane two pixels beyond the tile: the region carries n + 2 halo rows (and HALO >= n + 2 columns) for the n steps, so the final
// values are exact on the tile + 2; they go into an LDS window and akz_prep_passes.hpp -- the same four passes as k_prep's --
// writes the next level's Lsmooth and Lflow.  The host uses it only where no tile has a single in-image row or column
// ((w - 1) % TW != 0, (h - 1) % TH != 0: the passes' clamped reads then stay inside the window).
struct FedEpi {
    float* lsmooth;      // of the NEXT level
    float* lflow;
    PrepTaps taps;
    const double* d_k;   // contrast factors of octave 0, per image
    unsigned k_pow;      // octave of the next level (= this one's)
};
template <int TW, int TH, int HALO, int NT, bool EPI = false>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(AKZ_FED_WAVES, AKZ_FED_WAVES)))
k_fed_own(const float* __restrict__ L_in, const float* __restrict__ C, float* __restrict__ L_out,
          float* __restrict__ Lstep, int w, int h, FedTaus ht, FedEpi ep) {
    constexpr int RW = TW + 2 * HALO;    // region width in pixels (multiple of 4)
    constexpr int XG = RW / 4;           // float4 group columns
    constexpr int RP = RW + 8;           // LDS pitch: 4 pad floats on each side
    constexpr int RHMAX = TH + 2 * HALO;
    constexpr int PLANE = (RHMAX + 2) * RP;  // one pad row above and below
    static_assert(XG * (RHMAX / 2) <= NT, "one thread per pair of region rows and group column");
    static_assert(PLANE % 4 == 0, "the second plane stays 16-byte aligned");
    __shared__ __attribute__((aligned(16))) float sAB[2 * PLANE];
    float* const sA = sAB;
    float* const sB = sAB + PLANE;
    const int tid = threadIdx.x;
    const int steps = ht.n;
    const int n = steps + (EPI ? 2 : 0);  // halo rows: the steps, plus two for the epilogue's window (host: n <= HALO)
    const int RH = TH + 2 * n;  // even
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    const bool vec_ok = (w & 3) == 0;
    auto lds = [&](int ly, int lx) { return (ly + 1) * RP + 4 + lx; };

    // ---- stage Lt -> sA, Lflow -> sB (out-of-image pixels read as 0 and are never used) ----
    // The region has at most 2 * NT float4 groups per plane.  All four loads of a thread (two groups x two planes) are
    // issued before the first LDS write: a workgroup's lifetime is dominated by this load phase (one fused step costs
    // 17 us per full-resolution batch level, the launch 280), and a rolled loop would wait for the first pair of loads
    // before issuing the second.
    static_assert(RHMAX * XG <= 2 * NT && (NT - 1) / XG <= RHMAX, "two staging slots per thread, both inside the LDS planes");
    if (vec_ok) {
        const int i1 = tid + NT;
        const int ly0 = tid / XG, g0 = tid - ly0 * XG;
        const int ly1 = i1 / XG, g1 = i1 - ly1 * XG;
        const int gy0 = y0 - n + ly0, gx0 = x0 - HALO + 4 * g0;
        const int gy1 = y0 - n + ly1, gx1 = x0 - HALO + 4 * g1;
        // with w a multiple of 4 a group is either inside or outside the image as a whole
        const bool in0 = gy0 >= 0 && gy0 < h && gx0 >= 0 && gx0 < w;
        const bool in1 = i1 < RH * XG && gy1 >= 0 && gy1 < h && gx1 >= 0 && gx1 < w;
        const size_t o0 = in0 ? base + (size_t)gy0 * w + gx0 : base;
        const size_t o1 = in1 ? base + (size_t)gy1 * w + gx1 : base;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 l0 = *reinterpret_cast<const float4*>(L_in + o0), c0 = *reinterpret_cast<const float4*>(C + o0);
        float4 l1 = *reinterpret_cast<const float4*>(L_in + o1), c1 = *reinterpret_cast<const float4*>(C + o1);
        if (!in0) { l0 = zero; c0 = zero; }
        if (!in1) { l1 = zero; c1 = zero; }
        *reinterpret_cast<float4*>(&sA[lds(ly0, 4 * g0)]) = l0;
        *reinterpret_cast<float4*>(&sB[lds(ly0, 4 * g0)]) = c0;
        if (i1 < RH * XG) {
            *reinterpret_cast<float4*>(&sA[lds(ly1, 4 * g1)]) = l1;
            *reinterpret_cast<float4*>(&sB[lds(ly1, 4 * g1)]) = c1;
        }
    } else {
        for (int idx = tid; idx < RH * XG; idx += NT) {
            const int ly = idx / XG, g = idx - ly * XG;
            const int gy = y0 - n + ly, gx = x0 - HALO + 4 * g;
            float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
            if (gy >= 0 && gy < h) {
                const float* pl = L_in + base + (size_t)gy * w;
                const float* pc = C + base + (size_t)gy * w;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gx + e >= 0 && gx + e < w) {
                        a[e] = pl[gx + e];
                        b[e] = pc[gx + e];
                    }
            }
            *reinterpret_cast<float4*>(&sA[lds(ly, 4 * g)]) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4*>(&sB[lds(ly, 4 * g)]) = make_float4(b[0], b[1], b[2], b[3]);
        }
    }
    __syncthreads();

    // ---- ownership: group column g, region rows 2p and 2p+1 ----
    const int slot = tid < XG * (RH / 2) ? tid : 0;
    const bool active = tid < XG * (RH / 2);
    const int p = slot / XG, g = slot - p * XG;
    const int lya = 2 * p, lyb = 2 * p + 1;
    const int oa = lds(lya, 4 * g), ob = oa + RP;
    const int gx = x0 - HALO + 4 * g, gya = y0 - n + lya, gyb = gya + 1;
    const unsigned lane = threadIdx.x & 63u;

    // Row-pair vectors: .x belongs to owned row a, .y to owned row b, so that every flux and update below is one
    // packed f32 operation (v_pk_add_f32 / v_pk_mul_f32: both rows per instruction, IEEE per component, no FMA).
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 L[4];                 // Lt of the two owned rows
    v2 SX[5], SU[4], SV[4];  // Lflow pair sums, constant over the steps: x-direction | (north of a, a|b) | (a|b, south of b)
    {
        const float4 va = *reinterpret_cast<const float4*>(sA + oa), vb = *reinterpret_cast<const float4*>(sA + ob);
        L[0] = v2{va.x, vb.x}; L[1] = v2{va.y, vb.y}; L[2] = v2{va.z, vb.z}; L[3] = v2{va.w, vb.w};
        const float4 ca4 = *reinterpret_cast<const float4*>(sB + oa), cb4 = *reinterpret_cast<const float4*>(sB + ob);
        const float4 cn4 = *reinterpret_cast<const float4*>(sB + oa - RP);
        const float4 cs4 = *reinterpret_cast<const float4*>(sB + ob + RP);
        const float ca[6] = {sB[oa - 1], ca4.x, ca4.y, ca4.z, ca4.w, sB[oa + 4]};
        const float cb[6] = {sB[ob - 1], cb4.x, cb4.y, cb4.z, cb4.w, sB[ob + 4]};
        const float cn[4] = {cn4.x, cn4.y, cn4.z, cn4.w}, cs[4] = {cs4.x, cs4.y, cs4.z, cs4.w};
#pragma unroll
        for (int i = 0; i < 5; ++i) SX[i] = v2{ca[i] + ca[i + 1], cb[i] + cb[i + 1]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sN = cn[i] + ca[i + 1], sM = ca[i + 1] + cb[i + 1], sS = cb[i + 1] + cs[i];
            SU[i] = v2{sN, sM};
            SV[i] = v2{sM, sS};
        }
    }
    __syncthreads();  // sB is free from here on: it becomes the second Lt buffer

    // The image's border WITHOUT the case-by-case expressions of :84-137 (column 0 is a group's first; column w-1 may be any of a
    // group's four).  Every case is the full
    // expression  ((x_pos - x_neg) + y_pos) - y_neg  with the missing term replaced by a signed zero that makes the sum
    // bit-identical to the reference's shorter one:
    //   no left neighbour   t = x_pos            == x_pos - (+0)          x - (+0) == x for every x, -0 included
    //   no right neighbour  t = -x_neg           == (-0) - x_neg          (-0) - x == -x for every x, both zeros included
    //   first row           t = .. + y_pos       == (.. + y_pos) - (+0)
    //   last row            t = .. + c' (L(y-1) - L)  [:104-119: y_pos taken towards y-1, nothing subtracted]
    //                                            == (.. + y_pos') - (+0)  with y_pos' that very product: the pair sum and the
    //                                               neighbour of y_neg in y_pos's place
    // so a border thread runs the same packed instructions as every other plus a few selects, behind branches that are
    // uniform over the wave.  (Before, a thread with any border or outside pixel ran all eight of its pixels through the cases:
    // in the tiles on the image's left and right edges that is every wave, and a fused step of a small level -- whose launch
    // lasts as long as its slowest workgroup -- cost 0.85 us instead of 0.47: a third of a lone frame's coarse octaves.)
    const bool left = gx == 0;
    const int kr = w - gx;  // the flux to the right of the image's last column is XF[kr], if this group holds that column
    const bool right = kr >= 1 && kr <= 4;
    const bool zu_a = gya == 0 || gya == h - 1, zu_b = gyb == 0 || gyb == h - 1;  // no y_neg term: first row, last row
    const bool bot_a = gya == h - 1, bot_b = gyb == h - 1;
    const bool colcase = __ballot(left || right) != 0ull, rowcase = __ballot(zu_a || zu_b) != 0ull;  // (wave-uniform)
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // last row: y_pos' uses y_neg's pair sum
        if (bot_a) SV[i].x = SU[i].x;
        if (bot_b) SV[i].y = SU[i].y;
    }

    float* src = sA;
    float* dst = sB;
    v2 ST[4] = {v2{0.f, 0.f}, v2{0.f, 0.f}, v2{0.f, 0.f}, v2{0.f, 0.f}};
    for (int s = 1; s <= steps; ++s) {
        const float half_tau = ht.half_tau[s - 1];
        // left / right neighbours from the adjacent lanes (same row pair, neighbouring group column);
        // the first / last lane of a wave and region edges fall back to LDS (edge values are never used)
        v2 Lw = v2{from_left_lane(L[3].x), from_left_lane(L[3].y)};
        v2 Le = v2{from_right_lane(L[0].x), from_right_lane(L[0].y)};
        if (lane == 0u || g == 0) Lw = v2{src[oa - 1], src[ob - 1]};
        if (lane == 63u || g == XG - 1) Le = v2{src[oa + 4], src[ob + 4]};
        const float4 n4 = *reinterpret_cast<const float4*>(src + oa - RP);
        const float4 s4 = *reinterpret_cast<const float4*>(src + ob + RP);
        const float ln[4] = {n4.x, n4.y, n4.z, n4.w}, ls[4] = {s4.x, s4.y, s4.z, s4.w};
        const v2 R[6] = {Lw, L[0], L[1], L[2], L[3], Le};
        v2 XF[5];  // x fluxes between columns i-1 and i of the group (x_neg(x) and x_pos(x-1) are one expression, :63-64)
#pragma unroll
        for (int i = 0; i < 5; ++i) XF[i] = SX[i] * (R[i + 1] - R[i]);
        v2 U[4], V[4];  // (y_neg of a, y_pos of a == y_neg of b) and (y_pos of a, y_pos of b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            U[i] = SU[i] * (L[i] - v2{ln[i], L[i].x});
            V[i] = SV[i] * (v2{L[i].y, ls[i]} - L[i]);
        }
        if (colcase) {
            if (left) XF[0] = v2{0.0f, 0.0f};
#pragma unroll
            for (int k = 1; k <= 4; ++k)
                if (kr == k) XF[k] = v2{-0.0f, -0.0f};
        }
        if (rowcase) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (last row: the product c' (L(y-1) - L) in y_pos's place; row b's y-1 is row a)
                if (bot_a) V[i].x = SV[i].x * (ln[i] - L[i].x);
                if (bot_b) V[i].y = SV[i].y * (L[i].x - L[i].y);
                if (zu_a) U[i].x = 0.0f;
                if (zu_b) U[i].y = 0.0f;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ST[i] = half_tau * (((XF[i + 1] - XF[i]) + V[i]) - U[i]);
            L[i] = L[i] + ST[i];
        }
        if (s < steps) {
            if (active) {
                *reinterpret_cast<float4*>(dst + oa) = make_float4(L[0].x, L[1].x, L[2].x, L[3].x);
                *reinterpret_cast<float4*>(dst + ob) = make_float4(L[0].y, L[1].y, L[2].y, L[3].y);
            }
            __syncthreads();
            float* t = src;
            src = dst;
            dst = t;
        }
    }
    const float la[4] = {L[0].x, L[1].x, L[2].x, L[3].x}, lb[4] = {L[0].y, L[1].y, L[2].y, L[3].y};
    const float sta[4] = {ST[0].x, ST[1].x, ST[2].x, ST[3].x}, stb[4] = {ST[0].y, ST[1].y, ST[2].y, ST[3].y};

    // ---- centre groups go straight to HBM ----
    if (active && gx >= x0 && gx < x0 + TW) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int ly = r == 0 ? lya : lyb, gy = r == 0 ? gya : gyb;
            if (ly < n || ly >= n + TH || gy >= h || gx >= w) continue;
            const float* v = r == 0 ? la : lb;
            const float* st = r == 0 ? sta : stb;
            float* po = L_out + base + (size_t)gy * w;
            float* ps = Lstep ? Lstep + base + (size_t)gy * w : nullptr;
            if (vec_ok && gx + 3 < w) {
                plane_store4(po + gx, v[0], v[1], v[2], v[3]);
                if (ps) plane_store4(ps + gx, st[0], st[1], st[2], st[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (gx + e < w) {
                        plane_store(po + gx + e, v[e]);
                        if (ps) plane_store(ps + gx + e, st[e]);
                    }
            }
        }
    }
    if constexpr (EPI) {
        // ---- the next level's preparation from the final values on the tile + 2 ----
        constexpr int IW = TW + 4, IH = TH + 4;
        static_assert(IH * IW + (TH + 4) * (TW + 2) + (TH + 2) * (TW + 2) <= 2 * PLANE, "the preparation's windows fit the two planes");
        float* const wI = sAB;                        // input window, origin (x0 - 2, y0 - 2)
        float* const wA = wI + IH * IW;
        float* const wB = wA + (TH + 4) * (TW + 2);
        __syncthreads();  // (the last step's neighbour reads of the planes are done)
        if (active) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int gy = r == 0 ? gya : gyb;
                const int wy = gy - (y0 - 2);
                if (wy < 0 || wy >= IH || gy < 0 || gy >= h) continue;
                const float* v = r == 0 ? la : lb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int wx = gx + e - (x0 - 2);
                    if (wx >= 0 && wx < IW && gx + e >= 0 && gx + e < w) wI[wy * IW + wx] = v[e];
                }
            }
        }
        __syncthreads();
        const double kc = octave_contrast(ep.d_k[blockIdx.z], ep.k_pow);
        prep_passes<TW, TH, NT>(wI, wA, wB, x0, y0, w, h, base, ep.lsmooth, ep.lflow, ep.taps, 1.0 / (kc * kc));
    }
}

// ---------------------------------------------------------------------------------------------
// contrast factor (akaze/src/ops/contrast_factor.rs:18-71) on the sigma-blurred level 0
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double grad_mod(const float* I, int w, int h, int x, int y, Scharr1 sk) {
    float lx, ly;
    scharr1_pair(I, w, h, x, y, sk, lx, ly);
    const double dx = (double)lx, dy = (double)ly;
    return sqrt(dx * dx + dy * dy);
}
// Both passes walk the interior with a fixed number of fat workgroups per image (grid-stride over
// rows) so that each workgroup issues ONE global atomicMax / one histogram flush.
constexpr int CT = 256;  // threads per contrast workgroup
__global__ void __launch_bounds__(CT)
k_contrast_max(const float* __restrict__ blurred, int w, int h, Scharr1 sk,
               unsigned long long* __restrict__ d_hmax_bits) {
    __shared__ unsigned long long s_part[CT / 64];
    const float* I = blurred + (size_t)blockIdx.z * (size_t)w * (size_t)h;
    double m = 0.0;
    for (int y = 1 + (int)blockIdx.x; y < h - 1; y += (int)gridDim.x)
        for (int x = 1 + (int)threadIdx.x; x < w - 1; x += CT) {
            const double g = grad_mod(I, w, h, x, y, sk);
            if (g > m) m = g;
        }
    // non-negative doubles order like their bit patterns
    unsigned long long bits = (unsigned long long)__double_as_longlong(m);
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(bits, o, 64);
        bits = other > bits ? other : bits;
    }
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < CT / 64; ++i) bits = s_part[i] > bits ? s_part[i] : bits;
        if (bits != 0ull) atomicMax(d_hmax_bits + blockIdx.z, bits);
    }
}
__global__ void __launch_bounds__(CT)
k_contrast_hist(const float* __restrict__ blurred, int w, int h, Scharr1 sk,
                const unsigned long long* __restrict__ d_hmax_bits, unsigned nbins, unsigned copies,
                unsigned* __restrict__ d_hist) {
    extern __shared__ unsigned s_hist[];  // `copies` private histograms (lanes spread over them)
    const unsigned tid = threadIdx.x;
    for (unsigned b = tid; b < nbins * copies; b += CT) s_hist[b] = 0u;
    __syncthreads();
    const float* I = blurred + (size_t)blockIdx.z * (size_t)w * (size_t)h;
    const double hmax = __longlong_as_double((long long)d_hmax_bits[blockIdx.z]);
    unsigned* mine = s_hist + (tid & (copies - 1u)) * nbins;
    for (int y = 1 + (int)blockIdx.x; y < h - 1; y += (int)gridDim.x)
        for (int x = 1 + (int)tid; x < w - 1; x += CT) {
            const double g = grad_mod(I, w, h, x, y, sk);
            if (g != 0.0) {
                const double f = floor((double)nbins * (g / hmax));
                const unsigned b = f >= (double)nbins ? nbins - 1u : (f > 0.0 ? (unsigned)f : 0u);
                atomicAdd(&mine[b], 1u);
            }
        }
    __syncthreads();
    for (unsigned b = tid; b < nbins; b += CT) {
        unsigned v = 0;
        for (unsigned c = 0; c < copies; ++c) v += s_hist[c * nbins + b];
        if (v) atomicAdd(&d_hist[(size_t)blockIdx.z * nbins + b], v);
    }
}
// One workgroup per image: the histogram is staged in LDS by all threads, then thread 0 replays the sequential
// percentile scan of contrast_factor.rs:56-70 (a single thread walking the bins in global memory took 38 us).
__global__ void __launch_bounds__(256)
k_contrast_final(const unsigned long long* __restrict__ d_hmax_bits, const unsigned* __restrict__ d_hist,
                 unsigned nbins, double percentile, unsigned n, double* __restrict__ d_k) {
    extern __shared__ unsigned s_bins[];
    const unsigned img = blockIdx.x;
    if (img >= n) return;
    const unsigned* hist = d_hist + (size_t)img * nbins;
    for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) s_bins[b] = hist[b];
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double hmax = __longlong_as_double((long long)d_hmax_bits[img]);
    unsigned long long num_points = 0;
    for (unsigned b = 0; b < nbins; ++b) num_points += s_bins[b];
    const double tf = (double)num_points * percentile;
    const unsigned long long threshold = tf > 0.0 ? (unsigned long long)tf : 0ull;
    unsigned long long k = 0, num_elements = 0;
    while (num_elements < threshold && k < nbins) {
        num_elements += s_bins[k];
        k += 1;
    }
    d_k[img] = num_elements >= threshold ? hmax * (double)k / (double)nbins : 0.03;
}

// The histogram pass behind k_head (akz_stencil.hip), and the percentile with it.  k_head has stored the Scharr pair of every
// pixel: this pass is elementwise -- four pixels per thread and trip, no neighbourhood -- over the interior (contrast_factor.rs:
// 27-38 walks x in 1..w-1, y in 1..h-1; the border pixels of the planes hold copies of interior values and are skipped).
// d_smax_bits holds the largest SQUARED gradient magnitude (hmax is its square root).  The workgroup that finishes an image
// last -- a ticket per workgroup, taken after its bins have been added -- replays the percentile scan (k_contrast_final) and
// leaves the bins, the maximum and the tickets ZERO for the next job: no launch for the scan, none for clearing the scratch.
template <int NTH>
__global__ void __launch_bounds__(NTH)
k_contrast_hist_final(const float* __restrict__ gxp, const float* __restrict__ gyp, int w, int h, unsigned long long* __restrict__ d_smax_bits,
                      unsigned nbins, unsigned copies, unsigned* __restrict__ d_hist, unsigned* __restrict__ d_done, double percentile,
                      double* __restrict__ d_k) {
    extern __shared__ unsigned s_hist[];  // `copies` private histograms (lanes spread over them); max(nbins * copies, nbins) words
    __shared__ unsigned s_last;
    const unsigned tid = threadIdx.x, img = blockIdx.z;
    for (unsigned b = tid; b < nbins * copies; b += NTH) s_hist[b] = 0u;
    __syncthreads();
    const size_t px = (size_t)w * (size_t)h;
    const float* GX = gxp + (size_t)img * px;
    const float* GY = gyp + (size_t)img * px;
    const double hmax = sqrt(__longlong_as_double((long long)d_smax_bits[img]));
    unsigned* mine = s_hist + (tid & (copies - 1u)) * nbins;
    const bool vec = (px & 3u) == 0;  // every image's plane starts on a 16-byte boundary
    const size_t nquads = (px + 3) / 4;
    for (size_t q = (size_t)blockIdx.x * NTH + tid; q < nquads; q += (size_t)gridDim.x * NTH) {
        const size_t i0 = q * 4;
        float ax[4], ay[4];
        if (vec) {
            const float4 a = *reinterpret_cast<const float4*>(GX + i0), b = *reinterpret_cast<const float4*>(GY + i0);
            ax[0] = a.x; ax[1] = a.y; ax[2] = a.z; ax[3] = a.w;
            ay[0] = b.x; ay[1] = b.y; ay[2] = b.z; ay[3] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ax[e] = i0 + e < px ? GX[i0 + e] : 0.0f;
                ay[e] = i0 + e < px ? GY[i0 + e] : 0.0f;
            }
        }
        int y = (int)(i0 / (size_t)w), x = (int)(i0 - (size_t)y * w);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
                const double dx = (double)ax[e], dy = (double)ay[e];
                const double g = sqrt(dx * dx + dy * dy);
                if (g != 0.0) {
                    const double f = floor((double)nbins * (g / hmax));
                    const unsigned b = f >= (double)nbins ? nbins - 1u : (f > 0.0 ? (unsigned)f : 0u);
                    atomicAdd(&mine[b], 1u);
                }
            }
            if (++x == w) {
                x = 0;
                ++y;
            }
        }
    }
    __syncthreads();
    unsigned* hist = d_hist + (size_t)img * nbins;
    for (unsigned b = tid; b < nbins; b += NTH) {
        unsigned v = 0;
        for (unsigned c = 0; c < copies; ++c) v += s_hist[c * nbins + b];
        if (v) {  // (the RETURNING form: the wave's wait below then covers the add's completion by construction)
            const unsigned before = atomicAdd(&hist[b], v);
            asm volatile("" ::"v"(before));
        }
    }
    // This workgroup's bins have been ADDED (device-scope atomics, performed where all XCDs see them) before its ticket is taken:
    // the wave waits for its outstanding memory operations, the barrier for all waves.  (Not __threadfence(): a device-scope
    // fence on this chip writes the XCD's L2 back -- 16 MB of k_head's planes, by every one of 512 workgroups: 51 us for this
    // kernel instead of 23.  Nothing but atomics is exchanged here: the bins -- added in the returning form, so that the wait
    // covers them by construction --, the ticket, and the last workgroup's atomic loads.)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&d_done[img], 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    for (unsigned b = tid; b < nbins; b += NTH) {
        s_hist[b] = __hip_atomic_load(&hist[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hist[b] = 0u;
    }
    __syncthreads();
    if (tid != 0) return;
    unsigned long long num_points = 0;
    for (unsigned b = 0; b < nbins; ++b) num_points += s_hist[b];
    const double tf = (double)num_points * percentile;
    const unsigned long long threshold = tf > 0.0 ? (unsigned long long)tf : 0ull;
    unsigned long long k = 0, num_elements = 0;
    while (num_elements < threshold && k < nbins) {
        num_elements += s_hist[k];
        k += 1;
    }
    d_k[img] = num_elements >= threshold ? hmax * (double)k / (double)nbins : 0.03;
    d_smax_bits[img] = 0ull;
    d_done[img] = 0u;
}
// Lflow = pm_g2 of a stored Scharr pair (lib.rs:98-105; k_head's pair is level 1's: its Lsmooth is the blur the contrast
// factor took of Lt0), elementwise
__global__ void __launch_bounds__(256)
k_flow_pair(const float* __restrict__ gxp, const float* __restrict__ gyp, float* __restrict__ lflow, size_t px, const double* __restrict__ d_k,
            unsigned k_pow) {
    const unsigned img = blockIdx.z;
    const double kc = octave_contrast(d_k[img], k_pow);
    const double inverse_k = 1.0 / (kc * kc);
    const float* GX = gxp + (size_t)img * px;
    const float* GY = gyp + (size_t)img * px;
    float* out = lflow + (size_t)img * px;
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= px) return;
    if ((px & 3u) == 0) {
        const float4 a = *reinterpret_cast<const float4*>(GX + i0), b = *reinterpret_cast<const float4*>(GY + i0);
        float4 o;
        o.x = pm_g2_px(a.x, b.x, inverse_k);
        o.y = pm_g2_px(a.y, b.y, inverse_k);
        o.z = pm_g2_px(a.z, b.z, inverse_k);
        o.w = pm_g2_px(a.w, b.w, inverse_k);
        *reinterpret_cast<float4*>(out + i0) = o;
    } else {
        for (size_t i = i0; i < i0 + 4 && i < px; ++i) out[i] = pm_g2_px(GX[i], GY[i], inverse_k);
    }
}

// Ldet = ((Lxx*Lyy) - (Lxy*Lxy)) * sigma^4   (akaze/src/ops/detector_response.rs:52)
__global__ void k_ldet(const float* __restrict__ lxx, const float* __restrict__ lyy, const float* __restrict__ lxy,
                       float* __restrict__ out, size_t count, float q) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    out[i] = ((lxx[i] * lyy[i]) - (lxy[i] * lxy[i])) * q;
}

// image_1[i] += image_2[i]: what the reference's `sqrt_squared` does despite its name (akaze/src/types/image.rs:218-231);
// the two pointers may be the same plane (scharr with both orders adds the horizontal derivative to itself)
__global__ void k_accumulate(float* a, const float* b, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    a[i] = a[i] + b[i];
}

// ---------------------------------------------------------------------------------------------
// NMS candidates (akaze/src/ops/scale_space_extrema.rs:32-42) pre-filtered by the descriptor
// border test (:80-87), which depends only on (x, y, level); out-of-border candidates never
// touch the reference's keypoint cache.  Unordered append; the host sorts into raster order.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void nms_emit(float v, float xp, float xm, float yp, float ym, int x, int y, int w, int h,
                                         unsigned level, unsigned img, float thr, float border_m,
                                         Candidate* __restrict__ cand, unsigned cap, unsigned* __restrict__ count) {
    if (!(v > thr)) return;
    if (!(v > xp && v > xm && v > ym && v > yp)) return;
    const float fx = (float)x, fy = (float)y;
    const bool is_out = (roundf(fx - border_m) - 1.0f) < 0.0f || (roundf(fx + border_m) + 1.0f) >= (float)w ||
                        (roundf(fy - border_m) - 1.0f) < 0.0f || (roundf(fy + border_m) + 1.0f) >= (float)h;
    if (is_out) return;
    const unsigned slot = atomicAdd(count, 1u);
    if (slot < cap) {
        Candidate c;
        c.level = level;
        c.idx = (unsigned)(y * w + x);
        c.v = v; c.xp = xp; c.xm = xm; c.yp = yp; c.ym = ym;
        c.img = img;
        cand[slot] = c;
    }
}
// one thread = 4 consecutive pixels of a row (float4 loads of the row and its two neighbours)
__global__ void k_nms(const float* __restrict__ ldet, int w, int h, size_t img_stride, unsigned level, float thr,
                      float border_m, Candidate* __restrict__ cand, unsigned cap, unsigned* __restrict__ count) {
    const int x0 = (blockIdx.x * BX + threadIdx.x) * 4, y = blockIdx.y * BY + threadIdx.y;
    if (x0 >= w || y < 1 || y >= h - 1) return;
    const float* D = ldet + (size_t)blockIdx.z * img_stride;
    const float* row = D + (size_t)y * w;
    float c[6], up[4], dn[4];
    if ((w & 3) == 0 && x0 + 3 < w) {
        const float4 vc = *reinterpret_cast<const float4*>(row + x0);
        const float4 vu = *reinterpret_cast<const float4*>(row - w + x0);
        const float4 vd = *reinterpret_cast<const float4*>(row + w + x0);
        c[1] = vc.x; c[2] = vc.y; c[3] = vc.z; c[4] = vc.w;
        up[0] = vu.x; up[1] = vu.y; up[2] = vu.z; up[3] = vu.w;
        dn[0] = vd.x; dn[1] = vd.y; dn[2] = vd.z; dn[3] = vd.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool in = x0 + e < w;
            c[e + 1] = in ? row[x0 + e] : 0.0f;
            up[e] = in ? row[x0 + e - w] : 0.0f;
            dn[e] = in ? row[x0 + e + w] : 0.0f;
        }
    }
    c[0] = x0 > 0 ? row[x0 - 1] : 0.0f;
    // the reference's "east" neighbour of x = w-1 is the next row's x = 0 (flat iteration); such a pixel
    // always fails the border test, so any value works; read in bounds.
    c[5] = (x0 + 4 < w) ? row[x0 + 4] : 0.0f;
    const long last = (long)w * h - w - 1;  // flat range (w+1) .. len-w-1 of the reference loop
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int x = x0 + e;
        if (x < 1 || x >= w) continue;
        if ((long)y * w + x >= last) continue;
        nms_emit(c[e + 1], c[e + 2], c[e], dn[e], up[e], x, y, w, h, level, blockIdx.z, thr, border_m, cand, cap, count);
    }
}

// published 7x7 half-Gaussian table (sigma 2.5) of SURF/KAZE/A-KAZE
// (akaze/src/ops/scale_space_extrema.rs:207-271)
__constant__ float c_gauss25[7][7] = {
    {0.02546481f, 0.02350698f, 0.01849125f, 0.01239505f, 0.00708017f, 0.00344629f, 0.00142946f},
    {0.02350698f, 0.02169968f, 0.01706957f, 0.01144208f, 0.00653582f, 0.00318132f, 0.00131956f},
    {0.01849125f, 0.01706957f, 0.01342740f, 0.00900066f, 0.00514126f, 0.00250252f, 0.00103800f},
    {0.01239505f, 0.01144208f, 0.00900066f, 0.00603332f, 0.00344629f, 0.00167749f, 0.00069579f},
    {0.00708017f, 0.00653582f, 0.00514126f, 0.00344629f, 0.00196855f, 0.00095820f, 0.00039744f},
    {0.00344629f, 0.00318132f, 0.00250252f, 0.00167749f, 0.00095820f, 0.00046640f, 0.00019346f},
    {0.00142946f, 0.00131956f, 0.00103800f, 0.00069579f, 0.00039744f, 0.00019346f, 0.00008024f}};

// ---------------------------------------------------------------------------------------------
// Dominant orientation, device part (akaze/src/ops/scale_space_extrema.rs:274-329).
// angs[k] = atan2(res_y, res_y) is pi/4 for res_y > 0 and <= 0 otherwise, so a window either
// takes every sample with res_y > 0 (in k order) or none; which windows do is a keypoint-
// independent bit mask computed on the host with the host libm.  The running sums are never
// reset (as in the reference).  The final atan2f is left to the host.
// ---------------------------------------------------------------------------------------------
// All images in one launch; the kernel (below) gathers the 109 weighted samples of 32 keypoints into LDS
// and then replays the sequential x / y running sums (f32 adds in k order, as the reference), two lanes
// per keypoint, exchanging the pair once per window.
struct OriTable {
    signed char a[112], b[112];
};
constexpr OriTable make_ori_table() {
    OriTable t{};
    int idx = 0;
    for (int i = -6; i <= 6; ++i)
        for (int j = -6; j <= 6; ++j)
            if (i * i + j * j < 36) {
                t.a[idx] = (signed char)i;
                t.b[idx] = (signed char)j;
                ++idx;
            }
    for (; idx < 112; ++idx) { t.a[idx] = 0; t.b[idx] = 0; }
    return t;
}
__constant__ OriTable c_ori = make_ori_table();
// The same samples ordered by row (b), then column (a): the order in which phase 1 hands them to consecutive threads, so
// that neighbouring lanes gather neighbouring pixels of a row; order[i] is the sample's index in the reference's order.
struct OriOrder {
    unsigned char k[112];
};
constexpr OriOrder make_ori_order() {
    OriOrder o{};
    const OriTable t = make_ori_table();
    int n = 0;
    for (int b = -6; b <= 6; ++b)
        for (int a = -6; a <= 6; ++a)
            for (int k = 0; k < 109; ++k)
                if (t.a[k] == a && t.b[k] == b) o.k[n++] = (unsigned char)k;
    for (; n < 112; ++n) o.k[n] = 0;
    return o;
}
__constant__ OriOrder c_ori_rows = make_ori_order();

// Workgroups are numbered so that each XCD (workgroup id mod 8) walks a CONTIGUOUS range of the keypoint
// list: neighbours in the list are neighbours in the image, and their sample windows then share lines in
// that XCD's L2 instead of being fetched once per XCD.  nb must be a multiple of 8.
__device__ __forceinline__ unsigned xcd_contiguous_group(unsigned b, unsigned nb) {
    return (b & 7u) * (nb >> 3) + (b >> 3);
}

// 32 keypoints per workgroup.  Phase 1: all 256 threads gather the 32 x 109 weighted samples into LDS.
// Phase 2: the window sums are one strictly sequential chain of f32 adds per keypoint and component (the
// sums are never reset between windows, scale_space_extrema.rs:298-328), so the first wave runs the 64
// chains of the workgroup side by side -- lane 2j carries sum_x of keypoint j, lane 2j+1 its sum_y --
// instead of spending a whole wave on the two chains of one keypoint.
constexpr int ORI_KPB = 32, ORI_NS = 109, ORI_NT = 256;
__global__ void __launch_bounds__(ORI_NT)
k_orientation(LevelTable tab, const KpParam* __restrict__ kps, unsigned nkp, const unsigned* __restrict__ d_nkp,
              unsigned long long window_mask, unsigned n_windows, OrientOut* __restrict__ out, unsigned out_stride) {
    __shared__ float s_rx[ORI_KPB * ORI_NS];
    __shared__ float s_ry[ORI_KPB * ORI_NS + 16];
    if (d_nkp) nkp = min(nkp, *d_nkp);  // (the count of a selection that ran on the device; nkp: what the grid was sized for)
    const unsigned base = xcd_contiguous_group(blockIdx.x, gridDim.x) * ORI_KPB;
    if (base >= nkp) return;  // whole workgroup
    // 14 samples per thread in two batches of 7: the gathers of a batch (keypoint -> level table -> Lx, Ly: three
    // dependent loads each) are all in flight before the first one is consumed
    constexpr int ORI_IT = (ORI_KPB * ORI_NS + ORI_NT - 1) / ORI_NT, ORI_UB = 7;
    static_assert(ORI_IT % ORI_UB == 0, "whole batches");
    for (int it0 = 0; it0 < ORI_IT; it0 += ORI_UB) {
        float vx[ORI_UB], vy[ORI_UB], gw[ORI_UB];
        unsigned slot[ORI_UB];
#pragma unroll
        for (int u = 0; u < ORI_UB; ++u) {
            const unsigned idx = threadIdx.x + (unsigned)(it0 + u) * ORI_NT;
            const unsigned j = idx / ORI_NS, k = c_ori_rows.k[idx - j * ORI_NS];
            vx[u] = 0.0f; vy[u] = 0.0f; gw[u] = 0.0f;
            slot[u] = j * ORI_NS + k;
            if (idx < ORI_KPB * ORI_NS && base + j < nkp) {
                const KpParam kp = kps[base + j];
                const LevelPtrs lv = tab.lv[kp.level];
                const float* lx = lv.lx + (size_t)kp.img * lv.stride;
                const float* ly = lv.ly + (size_t)kp.img * lv.stride;
                const int a = c_ori.a[k], b = c_ori.b[k];
                const float fy = roundf(kp.yf + (float)b * kp.scale);
                const float fx = roundf(kp.xf + (float)a * kp.scale);
                const int iy = clampi(fy > 0.0f ? (int)fy : 0, 0, (int)lv.h - 1);
                const int ix = clampi(fx > 0.0f ? (int)fx : 0, 0, (int)lv.w - 1);
                gw[u] = c_gauss25[a < 0 ? -a : a][b < 0 ? -b : b];
                const size_t p = (size_t)iy * lv.w + ix;
                vx[u] = lx[p];
                vy[u] = ly[p];
            }
        }
#pragma unroll
        for (int u = 0; u < ORI_UB; ++u) {
            const unsigned idx = threadIdx.x + (unsigned)(it0 + u) * ORI_NT;
            const unsigned j = idx / ORI_NS;
            if (idx < ORI_KPB * ORI_NS) {
                const bool have = base + j < nkp;
                s_rx[slot[u]] = have ? gw[u] * vx[u] : 0.0f;
                s_ry[slot[u]] = have ? gw[u] * vy[u] : 0.0f;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const unsigned lane = threadIdx.x, j = lane >> 1;
    const bool is_y = (lane & 1u) != 0;
    const float* py = s_ry + j * ORI_NS;
    const float* pa = is_y ? py : s_rx + j * ORI_NS;
    // The samples of a keypoint are the same for every window: they are read from LDS once, a sample that the
    // reference skips (res_y <= 0, scale_space_extrema.rs:303) becomes -0.0f, and x + (-0.0f) == x bit for bit for
    // every x, so each window is one unconditional chain of 109 register adds in the reference's order (the LDS
    // reads inside the chain made the kernel 62 us long whatever the keypoint count).
    float t[ORI_NS];
#pragma unroll
    for (int k = 0; k < ORI_NS; ++k) t[k] = py[k] > 0.0f ? pa[k] : -0.0f;
    float sum = 0.0f, maxv = 0.0f, bx = 0.0f, by = 0.0f;
    unsigned found = 0;
    for (unsigned wdw = 0; wdw < n_windows; ++wdw) {
        if ((window_mask >> wdw) & 1ull) {
#pragma unroll
            for (int k = 0; k < ORI_NS; ++k) sum = sum + t[k];
        }
        const float other = __shfl_xor(sum, 1, 64);
        const float sx = is_y ? other : sum, sy = is_y ? sum : other;
        const float val = sx * sx + sy * sy;
        if (val > maxv) {
            maxv = val;
            bx = sx;
            by = sy;
            found = 1;
        }
    }
    if (!is_y && base + j < nkp) {
        OrientOut o;
        o.sum_x = bx; o.sum_y = by; o.found = found; o.angle_bits = 0;
        out[(size_t)(base + j) * out_stride] = o;
    }
}

// ---------------------------------------------------------------------------------------------
// M-LDB descriptor (akaze/src/ops/descriptors.rs:37-175): one wave per keypoint; lanes 0..28 own
// the 4+9+16 grid cells and sum their samples sequentially in the reference order (f32 adds are
// not associative); then all 64 lanes evaluate the 162*channels comparisons and ballot packs
// them LSB-first into a 64-byte row.
// ---------------------------------------------------------------------------------------------
// pair order of mldb_binary_comparisons (descriptors.rs:161-174): for i, for j > i, per grid; cell ids 0..28
struct PairTable {
    unsigned char a[162], b[162];
};
constexpr PairTable make_pairs() {
    PairTable t{};
    int p = 0;
    const int counts[3] = {4, 9, 16}, base[3] = {0, 4, 13};
    for (int g = 0; g < 3; ++g)
        for (int i = 0; i < counts[g]; ++i)
            for (int j = i + 1; j < counts[g]; ++j) {
                t.a[p] = (unsigned char)(base[g] + i);
                t.b[p] = (unsigned char)(base[g] + j);
                ++p;
            }
    return t;
}
__constant__ PairTable c_pairs = make_pairs();
// The 486 bits of a 3-channel row, in order: (channel << 10) | (cell a << 5) | cell b -- the comparison behind bit i, so
// that the lanes need neither the divisions of the general formula nor two dependent table look-ups per bit
struct BitTable3 {
    unsigned short e[512];
};
constexpr BitTable3 make_bits3() {
    BitTable3 t{};
    const PairTable p = make_pairs();
    const int npairs[3] = {6, 36, 120}, pbase[3] = {0, 6, 42};
    int b = 0;
    for (int g = 0; g < 3; ++g)
        for (int pos = 0; pos < 3; ++pos)
            for (int q = 0; q < npairs[g]; ++q)
                t.e[b++] = (unsigned short)((pos << 10) | (p.a[pbase[g] + q] << 5) | p.b[pbase[g] + q]);
    for (; b < 512; ++b) t.e[b] = 0;  // a cell against itself: the bit is 0
    return t;
}
__constant__ BitTable3 c_bits3 = make_bits3();

// Four keypoints (waves) per workgroup.  The three grids sample the same rotated lattice: offsets
// k, l in [-10, 10) for the 2x2 and 4x4 grids and [-10, 11) for the 3x3 grid, and a sample's
// coordinates depend only on (k, l).  So the 64 lanes of a wave first gather the 21 x 21 lattice
// once (coordinates, three plane reads, rotation) into LDS -- 441 gathers instead of the 1241 of a
// per-grid evaluation -- and then one lane per cell (4 + 9 + 16 = 29 lanes, all grids at once) adds
// its samples sequentially in the reference order (f32 adds are not associative), so the gathers
// are never on the serial path.
//
// Workgroup numbering: xcd_contiguous_group, as in k_orientation.
constexpr int MLDB_KPB = 4;
constexpr int MLDB_LAT = 21, MLDB_NS = MLDB_LAT * MLDB_LAT;
// The keypoint's orientation on the device (akz_libm.hpp: this machine's atan2f / cosf / sinf as IEEE arithmetic), for the
// job whose selection ran on the device: the angle from the orientation sums (scale_space_extrema.rs:326), then its cosine and
// sine (descriptors.rs:55-56).  Wave-uniform (one keypoint per wave); lane 0 leaves the angle in the keypoint's record for the
// host.  `fma`: which of glibc's two sinf / cosf builds this machine's libm runs.  An argument the device forms do not cover
// raises *flag: the host then redoes the job's angles and descriptors with its own libm.
struct DeviceAngles {
    OrientOut* sums;     // NULL: (cos, sin) come from the host (cosi)
    unsigned stride;     // in OrientOut units
    unsigned fma;
    unsigned* flag;
};
// The job that is waited for: what the host wants of a keypoint -- its 32-byte record (position, response, level, orientation
// sums, angle) and its descriptor row -- is ALSO stored straight into the host's pinned buffers by the wave that finishes the
// keypoint, and the selection's headers by the first workgroup, instead of three copies behind the kernel: a lone frame's call
// ended with 35-60 us of copy-engine work and the gaps between the copies.  (Stores to host memory are posted writes over the
// link: they drain while the other waves compute; the host reads after the stream's synchronisation, as it did the copies.)
struct MldbHostMirror {
    uint4* recs;        // host: 2 x uint4 per keypoint (NULL: no mirror)
    const uint4* hdr;   // device: hdr_vec x uint4 of selection headers ...
    uint4* hdr_host;    // ... and their place on the host
    unsigned hdr_vec;
    uint8_t* desc;      // host: 64-byte descriptor rows (NULL: the caller keeps them on the device)
};
__device__ __forceinline__ float2 device_angle(const DeviceAngles& da, unsigned kpi, bool write, const MldbHostMirror& hm) {
    OrientOut* o = da.sums + (size_t)kpi * da.stride;
    const float ang = o->found ? libm::atan2f_glibc(o->sum_y, o->sum_x) : 0.0f;
    bool bad = false;
    float co, si;
    if (da.fma) {
        co = libm::cosf_glibc<true>(ang, &bad);
        si = libm::sinf_glibc<true>(ang, &bad);
    } else {
        co = libm::cosf_glibc<false>(ang, &bad);
        si = libm::sinf_glibc<false>(ang, &bad);
    }
    if (write) {
        o->angle_bits = __float_as_uint(ang);
        if (bad && da.flag) atomicOr(da.flag, 1u);  // (the host also sees it in the angle itself: NaN or |angle| >= 120)
        if (hm.recs) {  // (stride 2: the sums are the second half of the keypoint's 32-byte record)
            const uint4* rec = reinterpret_cast<const uint4*>(da.sums) + (size_t)kpi * 2u - 1u;
            uint4 r0 = rec[0], r1 = rec[1];
            r1.w = __float_as_uint(ang);
            hm.recs[(size_t)kpi * 2u] = r0;
            hm.recs[(size_t)kpi * 2u + 1u] = r1;
        }
    }
    return float2{co, si};
}
__global__ void __launch_bounds__(64 * MLDB_KPB)
k_mldb(LevelTable tab, const KpParam* __restrict__ kps, const float2* __restrict__ cosi, unsigned nkp, const unsigned* __restrict__ d_nkp,
       DeviceAngles da, unsigned channels, uint8_t* __restrict__ desc64, unsigned first, MldbHostMirror hm) {
    __shared__ float s_win[MLDB_KPB][3][MLDB_NS + 7];
    __shared__ float s_val[MLDB_KPB][3][32];
    if (hm.recs && blockIdx.x == 0)  // (before anything can return: a job without keypoints still reports its headers)
        for (unsigned i = threadIdx.x; i < hm.hdr_vec; i += 64 * MLDB_KPB) hm.hdr_host[i] = hm.hdr[i];
    if (d_nkp) nkp = min(nkp, *d_nkp);  // (the count of a selection that ran on the device; nkp: the last keypoint the grid covers)
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned group0 = first + xcd_contiguous_group(blockIdx.x, gridDim.x) * MLDB_KPB;  // (first: a multiple of MLDB_KPB)
    const unsigned kpi = group0 + wv;
    if (group0 >= nkp) return;  // whole workgroup
    const bool live = kpi < nkp;
    KpParam kp = kps[live ? kpi : 0];
    const float2 cs = da.sums ? device_angle(da, live ? kpi : 0, live && lane == 0, hm) : cosi[live ? kpi : 0];
    // one keypoint per wave: its level is wave-uniform, so the level's pointers come from the kernel arguments by
    // scalar loads (a per-lane index would send the whole table through scratch memory)
    const LevelPtrs lv = tab.lv[__builtin_amdgcn_readfirstlane(kp.level)];
    const size_t ioff = (size_t)kp.img * lv.stride;
    const float* Lt = lv.lt + ioff;
    const float* Lx = lv.lx + ioff;
    const float* Ly = lv.ly + ioff;
    const float co = cs.x, si = cs.y, scale = kp.scale;
    unsigned short bits3[8];  // this lane's comparisons (fetched before the gathers are waited for)
#pragma unroll
    for (int r = 0; r < 8; ++r) bits3[r] = c_bits3.e[r * 64 + (int)lane];
    if (live) {
        // All gathers of a lane (7 lattice points x 3 planes) are issued before the first one is consumed: with the
        // rolled loop a wave waited for seven memory round trips in a row (29 us per keypoint, 0.45 ms per batch).
        constexpr int NIT = (MLDB_NS + 63) / 64;
        const bool k_fast = fabsf(co) >= fabsf(si);  // wave-uniform
        unsigned pidx[NIT];
        int widx[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int sidx = min((int)lane + 64 * it, MLDB_NS - 1);  // the surplus lanes of the last round repeat a sample
            // consecutive lanes walk the lattice direction that is closer to the image's x axis (a step in k moves
            // (co, si) * scale pixels, a step in l (-si, co) * scale): their gathers then share cache lines
            const int qa = sidx / MLDB_LAT, qb = sidx - qa * MLDB_LAT;
            const int kk = k_fast ? qb : qa, ll = k_fast ? qa : qb;
            const int k = kk - 10, l = ll - 10;
            const float lf = (float)l + 0.5f, kf = (float)k + 0.5f;
            const float sample_y = kp.yf + (lf * co * scale + kf * si * scale);
            const float sample_x = kp.xf + (-lf * si * scale + kf * co * scale);
            const int y1 = clampi((int)roundf(sample_y), 0, (int)lv.h - 1);
            const int x1 = clampi((int)roundf(sample_x), 0, (int)lv.w - 1);
            pidx[it] = (unsigned)y1 * (unsigned)lv.w + (unsigned)x1;
            widx[it] = kk * MLDB_LAT + ll;
        }
        float vt[NIT], vx[NIT], vy[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            vt[it] = Lt[pidx[it]];
            vx[it] = channels > 1 ? Lx[pidx[it]] : 0.0f;
            vy[it] = channels > 1 ? Ly[pidx[it]] : 0.0f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int sidx = (int)lane + 64 * it;
            if (sidx >= MLDB_NS) continue;
            float v1 = 0.0f, v2 = 0.0f;
            if (channels > 1) {
                const float rx = vx[it], ry = vy[it];
                if (channels == 2) {
                    v1 = sqrtf(rx * rx + ry * ry);
                } else {
                    v2 = rx * co + ry * si;   // rry -> dy
                    v1 = -rx * si + ry * co;  // rrx -> dx
                }
            }
            s_win[wv][0][widx[it]] = vt[it];
            s_win[wv][1][widx[it]] = v1;
            s_win[wv][2][widx[it]] = v2;
        }
    }
    __syncthreads();
    // Cell sums: lanes 0..28 of wave 0 take the cells of the workgroup's keypoint 0, lanes 32..60 those of keypoint 1;
    // wave 1 keypoints 2 and 3 the same way (a wave that sums for one keypoint issues the same instructions for 29 lanes)
    const unsigned cl = lane & 31u, ckp = 2u * wv + (lane >> 5);  // cell and keypoint (of the workgroup) of this lane
    if (wv < MLDB_KPB / 2 && cl < 29 && group0 + ckp < nkp) {
        // cell ids 0..3: 2x2 grid (step 10), 4..12: 3x3 (step 7), 13..28: 4x4 (step 5)
        const int g = cl < 4 ? 0 : (cl < 13 ? 1 : 2);
        const int step = g == 0 ? 10 : (g == 1 ? 7 : 5), ng = g + 2;
        const int ci = (int)cl - (g == 0 ? 0 : (g == 1 ? 4 : 13));
        const int kk0 = (ci / ng) * step, ll0 = (ci % ng) * step;  // lattice origin of the cell
        const int per_cell = step * step;
        const float* b0 = &s_win[ckp][0][0];
        const float* b1 = &s_win[ckp][1][0];
        const float* b2 = &s_win[ckp][2][0];
        float di = 0.0f, dx = 0.0f, dy = 0.0f;
        // k outer, l inner (descriptors.rs:108-109).  A lattice row of the cell is read at once (ten values per plane,
        // whatever the cell's width: the surplus ones lie in the next row or in the padding and are not added) and then
        // added in order: one LDS round trip per row instead of one per sample on the serial path; the loop is unrolled
        // over all ten possible rows (rows beyond the cell re-read its last row and add nothing) so that the reads of
        // the next rows are in flight while a row is added
#pragma unroll
        for (int dk = 0; dk < 10; ++dk) {
            const int o = (kk0 + min(dk, step - 1)) * MLDB_LAT + ll0;
            float r0[10], r1[10], r2[10];
#pragma unroll
            for (int dl = 0; dl < 10; ++dl) {
                r0[dl] = b0[o + dl];
                r1[dl] = b1[o + dl];
                r2[dl] = b2[o + dl];
            }
#pragma unroll
            for (int dl = 0; dl < 10; ++dl)
                if (dk < step && dl < step) {
                    di = di + r0[dl];
                    dx = dx + r1[dl];
                    dy = dy + r2[dl];
                }
        }
        const float ns = (float)per_cell;
        s_val[ckp][0][cl] = di / ns;
        s_val[ckp][1][cl] = dx / ns;
        s_val[ckp][2][cl] = dy / ns;
    }
    __syncthreads();
    // bit order: grid 0 (6 pairs), grid 1 (36), grid 2 (120); inside a grid channel-major
    const unsigned seg0 = 6u * channels, seg1 = seg0 + 36u * channels, total = seg1 + 120u * channels;
    unsigned long long words[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const unsigned b = (unsigned)r * 64u + lane;
        bool bit = false;
        if (channels == 3) {  // (uniform)
            const unsigned e = bits3[r];
            bit = s_val[wv][e >> 10][(e >> 5) & 31u] > s_val[wv][e >> 10][e & 31u];
        } else if (b < total) {
            unsigned rel, npairs, pbase;
            if (b < seg0) { rel = b; npairs = 6; pbase = 0; }
            else if (b < seg1) { rel = b - seg0; npairs = 36; pbase = 6; }
            else { rel = b - seg1; npairs = 120; pbase = 42; }
            const unsigned pos = rel / npairs, p = rel % npairs;
            bit = s_val[wv][pos][c_pairs.a[pbase + p]] > s_val[wv][pos][c_pairs.b[pbase + p]];
        }
        words[r] = __ballot(bit);
    }
    if (live && lane < 8) {
        unsigned long long wvw = words[0];
#pragma unroll
        for (int r = 1; r < 8; ++r)
            if (lane == (unsigned)r) wvw = words[r];
        reinterpret_cast<unsigned long long*>(desc64 + (size_t)kpi * 64)[lane] = wvw;
        if (hm.desc) reinterpret_cast<unsigned long long*>(hm.desc + (size_t)kpi * 64)[lane] = wvw;
    }
}

// ---------------------------------------------------------------------------------------------
// Brute-force Hamming 1-NN / 2-NN (akaze/src/ops/feature_matching.rs:37-50, :113-123).
// The reference's bail-out only skips distances that could not update min/second, so exact
// distances with the same update rule give identical (min, second, min_j).
// ---------------------------------------------------------------------------------------------
constexpr int MT = 256;  // queries per workgroup == train rows per LDS tile

// tail_mask: which bits of the last dword (bytes 60..63 of a row) are compared — 0xff for M-LDB rows (61 bytes;
// bytes 61..63 are padding that neither matcher kernel looks at), all ones for full 64-byte rows
__device__ __forceinline__ unsigned hamming64(const uint4 (&q)[4], const uint4* __restrict__ row, unsigned tail_mask) {
    unsigned d = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4 t = row[k];
        d += __popc(q[k].x ^ t.x);
        d += __popc(q[k].y ^ t.y);
        d += __popc(q[k].z ^ t.z);
        d += __popc((q[k].w ^ t.w) & (k == 3 ? tail_mask : 0xffffffffu));
    }
    return d;
}
__device__ __forceinline__ void top2_feed(unsigned d, unsigned j, unsigned& min_d, unsigned& second, unsigned& min_j) {
    if (d < min_d) {  // the reference's update rule (feature_matching.rs:41-49)
        second = min_d;
        min_d = d;
        min_j = j;
    } else if (d < second) {
        second = d;
    }
}
// blockIdx.x: 256 queries (one per thread, in registers); blockIdx.y: a chunk of the train set, staged
// through LDS 256 rows at a time.  Each workgroup writes the chunk-local (min, second, argmin) of its
// queries; k_match_compact folds the chunks in index order, which reproduces the sequential scan exactly
// (top-2 of a union = top-2 of the per-part top-2s; ties keep the lowest index).
__global__ void __launch_bounds__(MT) k_match(const uint4* __restrict__ d0, unsigned n0, const uint4* __restrict__ d1,
                                             unsigned n1, unsigned chunk_rows, unsigned threshold, unsigned tail_mask,
                                             MatchRec* __restrict__ out) {
    __shared__ uint4 s_tile[MT * 4];
    const unsigned i = blockIdx.x * MT + threadIdx.x;
    uint4 q[4];
    const bool live = i < n0;
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = live ? d0[(size_t)i * 4 + k] : make_uint4(0, 0, 0, 0);
    unsigned min_d = threshold, second = threshold, min_j = 0;
    const unsigned begin = blockIdx.y * chunk_rows, end = min(n1, begin + chunk_rows);
    for (unsigned base = begin; base < end; base += MT) {
        const unsigned rows = min((unsigned)MT, end - base);
        __syncthreads();
        for (unsigned e = threadIdx.x; e < rows * 4; e += MT) s_tile[e] = d1[(size_t)base * 4 + e];
        __syncthreads();
        unsigned r = 0;
        for (; r + 4 <= rows; r += 4) {  // four independent popcount chains per iteration
            const unsigned da = hamming64(q, s_tile + (r + 0) * 4, tail_mask), db = hamming64(q, s_tile + (r + 1) * 4, tail_mask);
            const unsigned dc = hamming64(q, s_tile + (r + 2) * 4, tail_mask), dd = hamming64(q, s_tile + (r + 3) * 4, tail_mask);
            top2_feed(da, base + r + 0, min_d, second, min_j);
            top2_feed(db, base + r + 1, min_d, second, min_j);
            top2_feed(dc, base + r + 2, min_d, second, min_j);
            top2_feed(dd, base + r + 3, min_d, second, min_j);
        }
        for (; r < rows; ++r) top2_feed(hamming64(q, s_tile + r * 4, tail_mask), base + r, min_d, second, min_j);
    }
    if (live) {
        MatchRec m;
        m.min_d = min_d; m.second_d = second; m.min_j = min_j; m._pad = 0;
        out[(size_t)blockIdx.y * n0 + i] = m;
    }
}
// Merge of every query's records over the chunks of one train set, one thread per query (the pair call has up to a
// hundred chunks: too many records for the single workgroup of k_match_compact to fold on its own).
__global__ void k_match_merge(const MatchRec* __restrict__ part, unsigned n0, unsigned chunks, unsigned threshold,
                              MatchRec* __restrict__ out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n0) return;
    unsigned min_d = threshold, second = threshold, min_j = 0;
    for (unsigned c = 0; c < chunks; ++c) {
        const MatchRec m = part[(size_t)c * n0 + i];
        top2_feed(m.min_d, m.min_j, min_d, second, min_j);
        if (m.second_d < second) second = m.second_d;  // second_d >= min_d of its chunk >= min_d
    }
    MatchRec o;
    o.min_d = min_d; o.second_d = second; o.min_j = min_j; o._pad = 0;
    out[i] = o;
}

// Merge of a query's records over the chunks of one train set (chunks in ascending row order: the strict '<' of
// top2_feed keeps the lowest row among equal minima, as the sequential scan does), then the Lowe ratio^2 + threshold
// test (feature_matching.rs:61-63) and ordered compaction; one workgroup per train set.
__global__ void __launch_bounds__(1024) k_match_compact(const MatchRec* __restrict__ rec, unsigned n0, unsigned chunks,
                                                        unsigned threshold, double ratio2,
                                                        akz_match* __restrict__ out,
                                                        unsigned long long* __restrict__ n_out) {
    // 16 x 1024 queries per round: every thread folds its 16 queries' chunk records (independent loads, issued back to
    // back), the waves publish their counts per group of 1024, ONE barrier, and every thread finds the place of its
    // matches from the 16 x 16 table -- two barriers per 16 K queries instead of three per 1 K.  (Many chunks are still
    // merged by k_match_merge first: this is one workgroup, and one compute unit's load path folds 11 chunks of 11 K
    // queries in 35 us.)
    constexpr int IT = 16;
    __shared__ unsigned s_cnt[IT][16];
    __shared__ unsigned s_base;
    rec += (size_t)blockIdx.x * chunks * n0;  // records [chunk][query] of this set (blockIdx.x = 0 for a single set)
    out += (size_t)blockIdx.x * n0;
    n_out += blockIdx.x;
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_base = 0;
    __syncthreads();
    for (unsigned start = 0; start < n0; start += IT * 1024u) {
        unsigned md[IT], mj[IT], keepmask = 0u;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const unsigned i = start + (unsigned)it * 1024u + threadIdx.x;
            bool keep = false;
            md[it] = mj[it] = 0u;
            if (i < n0) {
                unsigned min_d = threshold, second = threshold, min_j = 0;
                for (unsigned c = 0; c < chunks; ++c) {  // ascending rows: the strict '<' keeps the lowest row among equal minima
                    const MatchRec p = rec[(size_t)c * n0 + i];
                    top2_feed(p.min_d, p.min_j, min_d, second, min_j);
                    if (p.second_d < second) second = p.second_d;  // second_d >= min_d of its chunk >= min_d
                }
                md[it] = min_d; mj[it] = min_j;
                keep = ((double)min_d < (double)second * ratio2) && (min_d < threshold);
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
    if (threadIdx.x == 0) *n_out = s_base;
}

// Merge + ratio test + ORDERED compaction of one train set in ONE launch of many workgroups (the pair call: 11 chunks of
// 11 K queries took a 44-workgroup merge, 5 us, and then the single-workgroup compaction above, 9 us).  A workgroup folds
// the chunk records of its 256 queries exactly as k_match_merge does, counts its accepted matches and publishes the
// count in state[blockIdx.x] = epoch << 32 | count; its place in the output is the sum of the counts of the workgroups
// before it, which it reads as they appear (a look-back over aggregates: nobody waits for a prefix, only for counts that
// every workgroup publishes as soon as it has merged).  Workgroups are dispatched in index order and wait only for lower
// indices, so the wait terminates however many of them the chip holds at a time.  `epoch` differs from call to call: the
// state words of earlier calls read as "not yet published" and the buffer is never cleared.  Output order = query order,
// as feature_matching.rs:37-81 produces it.
__global__ void __launch_bounds__(256) k_match_merge_compact(const MatchRec* __restrict__ part, unsigned n0, unsigned chunks,
                                                             unsigned threshold, double ratio2, akz_match* __restrict__ out,
                                                             unsigned long long* __restrict__ n_out,
                                                             unsigned long long* __restrict__ state, unsigned epoch) {
    __shared__ unsigned s_cnt[4];
    __shared__ unsigned s_prefix;
    const unsigned b = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const unsigned i = b * 256u + tid;
    unsigned min_d = threshold, second = threshold, min_j = 0;
    bool keep = false;
    if (i < n0) {
        for (unsigned c = 0; c < chunks; ++c) {  // ascending rows: the strict '<' keeps the lowest row among equal minima
            const MatchRec m = part[(size_t)c * n0 + i];
            top2_feed(m.min_d, m.min_j, min_d, second, min_j);
            if (m.second_d < second) second = m.second_d;
        }
        keep = ((double)min_d < (double)second * ratio2) && (min_d < threshold);
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) s_cnt[wave] = (unsigned)__popcll(bal);
    __syncthreads();
    const unsigned total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    unsigned below = 0;
    for (unsigned w = 0; w < wave; ++w) below += s_cnt[w];
    if (tid == 0)
        __hip_atomic_store(state + b, ((unsigned long long)epoch << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave == 0) {  // the counts of the workgroups before this one, 64 at a time
        unsigned sum = 0;
        for (unsigned j0 = 0; j0 < b; j0 += 64u) {
            const unsigned j = j0 + lane;
            unsigned v = 0;
            if (j < b) {
                unsigned long long w;
                do {
                    w = __hip_atomic_load(state + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } while ((unsigned)(w >> 32) != epoch);
                v = (unsigned)w;
            }
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            sum += v;
        }
        if (lane == 0) s_prefix = sum;
    }
    __syncthreads();
    const unsigned prefix = s_prefix;
    if (keep) {
        akz_match o;
        o.index_0 = i; o.index_1 = min_j; o.distance = (double)min_d;
        out[prefix + below + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = o;
    }
    if (b == gridDim.x - 1 && tid == 0) *n_out = (unsigned long long)prefix + total;
}

inline dim3 grid2d(uint32_t w, uint32_t h, uint32_t n) { return dim3((w + BX - 1) / BX, (h + BY - 1) / BY, n); }

}  // namespace

// =============================================================================================
// launch wrappers
// =============================================================================================
namespace launch {

static Scharr1 scharr1() {
    const Taps m = taps_scharr_main(1);
    return Scharr1{m.wgt[0], m.wgt[1]};
}

void filter_h_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const Taps& t) {
    hipLaunchKernelGGL(k_filter_h<float>, grid2d(w, h, n), dim3(BX, BY), 0, s, in, out, (int)w, (int)h, t);
}
void filter_h_u8(hipStream_t s, const uint8_t* in, float* out, uint32_t w, uint32_t h, uint32_t n, const Taps& t) {
    hipLaunchKernelGGL(k_filter_h<uint8_t>, grid2d(w, h, n), dim3(BX, BY), 0, s, in, out, (int)w, (int)h, t);
}
void filter_v_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const Taps& t) {
    hipLaunchKernelGGL(k_filter_v, grid2d(w, h, n), dim3(BX, BY), 0, s, in, out, (int)w, (int)h, t);
}
void delay(hipStream_t s, uint32_t microseconds) {
    hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, s, (unsigned long long)microseconds * 100ull, (unsigned*)nullptr);
}
void half_size(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n) {
    if ((w & 7u) == 0 && (((uintptr_t)in | (uintptr_t)out) & 15u) == 0) {
        hipLaunchKernelGGL(k_half_size4, grid2d(w / 4, h / 2, n), dim3(BX, BY), 0, s, in, out, (int)w, (int)h);
        return;
    }
    hipLaunchKernelGGL(k_half_size, grid2d(w / 2, h / 2, n), dim3(BX, BY), 0, s, in, out, (int)w, (int)h);
}
void pm_g2(hipStream_t s, const float* lx, const float* ly, float* out, uint32_t w, uint32_t h, uint32_t n,
           const double* d_k, uint32_t k_scale_pow) {
    const size_t plane = (size_t)w * h;
    hipLaunchKernelGGL(k_pm_g2, dim3((unsigned)((plane + 255) / 256), 1, n), dim3(256), 0, s, lx, ly, out, plane, d_k,
                       k_scale_pow);
}
void flow(hipStream_t s, const float* lsmooth, float* lflow, uint32_t w, uint32_t h, uint32_t n, const double* d_k,
          uint32_t k_scale_pow) {
    hipLaunchKernelGGL(k_flow, grid2d(w, h, n), dim3(BX, BY), 0, s, lsmooth, lflow, (int)w, (int)h, scharr1(), d_k,
                       k_scale_pow);
}
void fed_step(hipStream_t s, const float* lt_in, const float* lflow, float* lt_out, float* lstep, uint32_t w,
              uint32_t h, uint32_t n, float half_tau) {
    hipLaunchKernelGGL(k_fed_step, grid2d(w, h, n), dim3(BX, BY), 0, s, lt_in, lflow, lt_out, lstep, (int)w, (int)h,
                       half_tau);
}
constexpr int kFedDeepTh = 10;
uint64_t fed_deep_workgroups(uint32_t w, uint32_t h, uint32_t n) {
    return (uint64_t)((w + 63) / 64) * ((h + kFedDeepTh - 1) / kFedDeepTh) * n;
}
// Can the next level's preparation ride on a diffusion launch of n_steps steps over w x h (k_fed_own<.., EPI>)?  The region
// needs n_steps + 2 halo rows and columns inside the template's HALO, and no tile may have a single in-image row or column.
bool fed_epilogue_supported(uint32_t w, uint32_t h, uint32_t n_steps) {
    if (n_steps == 0 || n_steps > 14 || w < 8 || h < 8 || (w - 1) % 64 == 0) return false;
    const uint32_t th = n_steps <= 6 ? 32u : (uint32_t)kFedDeepTh;
    return (h - 1) % th != 0;
}
void fed_fused(hipStream_t s, const float* lt_in, const float* lflow, float* lt_out, float* lstep, uint32_t w,
               uint32_t h, uint32_t n, const float* half_taus, uint32_t n_steps, const FedNextPrep* next) {
    constexpr int TW = 64, TH = 32, NT = 512;
    FedTaus ht;
    ht.n = (int)n_steps;
    for (uint32_t i = 0; i < 16; ++i) ht.half_tau[i] = i < n_steps ? half_taus[i] : 0.0f;
    const dim3 grid((w + TW - 1) / TW, (h + TH - 1) / TH, n);
    FedEpi ep{};
    if (next) {  // (the caller has asked fed_epilogue_supported)
        const Taps m = taps_scharr_main(1);
        ep.lsmooth = next->lsmooth;
        ep.lflow = next->lflow;
        ep.taps = PrepTaps{next->g3[0], next->g3[1], next->g3[2], m.wgt[0], m.wgt[1]};
        ep.d_k = next->d_k;
        ep.k_pow = next->k_pow;
        if (n_steps <= 6) {
            hipLaunchKernelGGL((k_fed_own<TW, TH, 8, NT, true>), grid, dim3(NT), 0, s, lt_in, lflow, lt_out, lstep, (int)w, (int)h, ht, ep);
        } else {
            const dim3 g10((w + TW - 1) / TW, (h + kFedDeepTh - 1) / kFedDeepTh, n);
            hipLaunchKernelGGL((k_fed_own<TW, kFedDeepTh, 16, NT, true>), g10, dim3(NT), 0, s, lt_in, lflow, lt_out, lstep, (int)w, (int)h, ht,
                               ep);
        }
        return;
    }
    {
        if (n_steps <= 4) {
            // halo 4: a 64 x 48 tile gives 18 x 27 (n = 3) or 18 x 28 (n = 4) owner threads of 512; 64 x 32 only 342 / 360
            // (64 x 40 and 256-thread 64 x 20 tiles measured slower)
            if (h >= 48) {
                const dim3 g48((w + TW - 1) / TW, (h + 47) / 48, n);
                hipLaunchKernelGGL((k_fed_own<TW, 48, 4, NT>), g48, dim3(NT), 0, s, lt_in, lflow, lt_out, lstep, (int)w,
                                   (int)h, ht, ep);
            } else
                hipLaunchKernelGGL((k_fed_own<TW, TH, 4, NT>), grid, dim3(NT), 0, s, lt_in, lflow, lt_out, lstep, (int)w,
                                   (int)h, ht, ep);
        } else if (n_steps <= 8)
            hipLaunchKernelGGL((k_fed_own<TW, TH, 8, NT>), grid, dim3(NT), 0, s, lt_in, lflow, lt_out, lstep, (int)w,
                               (int)h, ht, ep);
        else {
            // up to 16 steps on 64 x 10 tiles (region 96 x 42: 6.3 x the tile, so only where a launch is a handful of
            // workgroups at the floor of a dependent dispatch and halving the launches is what counts)
            const dim3 g10((w + TW - 1) / TW, (h + kFedDeepTh - 1) / kFedDeepTh, n);
            hipLaunchKernelGGL((k_fed_own<TW, kFedDeepTh, 16, NT>), g10, dim3(NT), 0, s, lt_in, lflow, lt_out, lstep,
                               (int)w, (int)h, ht, ep);
        }
    }
}
// Workgroups per image of the two contrast passes: 128 fat ones for a batch (one atomicMax / one histogram flush
// each); a small batch gets more, shorter ones so that the chip is filled (a lone 1080p frame: 38 + 36 us with 128).
#ifndef AKZ_HIST_BLOCKS
#define AKZ_HIST_BLOCKS 128u  // workgroups of the elementwise histogram pass (each flushes its bins once)
#endif
static unsigned contrast_blocks(uint32_t h, uint32_t n) {
    constexpr uint32_t target = 512u;  // lone 1080p frame: 37 + 34 us with 128, 18 + 24 with 512, 19 + 33 with 1024
    const uint32_t per_image = std::max<uint32_t>(128u, target / std::max<uint32_t>(n, 1u));
    return h > 2 ? std::min<uint32_t>(h - 2, per_image) : 1u;
}
void contrast_max(hipStream_t s, const float* blurred, uint32_t w, uint32_t h, uint32_t n,
                  unsigned long long* d_hmax_bits) {
    hipLaunchKernelGGL(k_contrast_max, dim3(contrast_blocks(h, n), 1, n), dim3(CT), 0, s, blurred, (int)w, (int)h,
                       scharr1(), d_hmax_bits);
}
void contrast_hist(hipStream_t s, const float* blurred, uint32_t w, uint32_t h, uint32_t n,
                   const unsigned long long* d_hmax_bits, uint32_t nbins, uint32_t* d_hist) {
    const unsigned copies = nbins <= 512 ? 8u : (nbins <= 2048 ? 2u : 1u);
    hipLaunchKernelGGL(k_contrast_hist, dim3(contrast_blocks(h, n), 1, n), dim3(CT), nbins * copies * sizeof(unsigned), s,
                       blurred, (int)w, (int)h, scharr1(), d_hmax_bits, nbins, copies, d_hist);
}
void contrast_final(hipStream_t s, const unsigned long long* d_hmax_bits, const uint32_t* d_hist, uint32_t nbins,
                    double percentile, uint32_t n, double* d_k) {
    hipLaunchKernelGGL(k_contrast_final, dim3(n), dim3(256), nbins * sizeof(unsigned), s, d_hmax_bits, d_hist, nbins, percentile,
                       n, d_k);
}
constexpr int HT = 1024;  // threads per workgroup of the elementwise histogram pass
void contrast_hist_final(hipStream_t s, const float* gx, const float* gy, uint32_t w, uint32_t h, uint32_t n, unsigned long long* d_smax_bits,
                         uint32_t nbins, uint32_t* d_hist, uint32_t* d_done, double percentile, double* d_k) {
    const unsigned copies = nbins <= 512 ? 8u : (nbins <= 2048 ? 2u : 1u);
    const uint64_t quads = ((uint64_t)w * h + 3) / 4;
    // 128 workgroups of 1 024 threads (each adds its bins to the image's once): 19.6 us for a lone 1080p frame; 512 x 256 threads
    // 25.7 -- the same threads, four times the flushes onto the same 300 counters
    const unsigned blocks = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((quads + HT - 1) / HT, std::max<uint32_t>(32u, AKZ_HIST_BLOCKS / std::max<uint32_t>(n, 1u))));
    hipLaunchKernelGGL(k_contrast_hist_final<HT>, dim3(blocks, 1, n), dim3(HT), nbins * copies * sizeof(unsigned), s, gx, gy, (int)w, (int)h,
                       d_smax_bits, nbins, copies, d_hist, d_done, percentile, d_k);
}
void flow_from_pair(hipStream_t s, const float* gx, const float* gy, float* lflow, uint32_t w, uint32_t h, uint32_t n, const double* d_k,
                    uint32_t k_pow) {
    const uint64_t px = (uint64_t)w * h;
    hipLaunchKernelGGL(k_flow_pair, dim3((unsigned)((px + 1023) / 1024), 1, n), dim3(256), 0, s, gx, gy, lflow, (size_t)px, d_k, k_pow);
}
void ldet(hipStream_t s, const float* lxx, const float* lyy, const float* lxy, float* out, uint64_t count,
          float sigma_quat) {
    hipLaunchKernelGGL(k_ldet, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, lxx, lyy, lxy, out,
                       (size_t)count, sigma_quat);
}
void rcp_f64_to_f32(hipStream_t s, const double* x, float* out, uint64_t n) {
    hipLaunchKernelGGL(k_rcp_f64_to_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, out, (size_t)n);
}
void accumulate(hipStream_t s, float* a, const float* b, uint64_t count) {
    hipLaunchKernelGGL(k_accumulate, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, a, b, (size_t)count);
}
void nms(hipStream_t s, const float* ldet_p, uint32_t w, uint32_t h, uint32_t n, uint64_t img_stride, uint32_t level,
         float thr, float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count) {
    hipLaunchKernelGGL(k_nms, grid2d((w + 3) / 4, h, n), dim3(BX, BY), 0, s, ldet_p, (int)w, (int)h, (size_t)img_stride,
                       level, thr, border_m, d_cand, cap, d_count);
}
void orientation(hipStream_t s, const LevelTable& lt, const KpParam* d_kp, uint32_t nkp,
                 unsigned long long window_mask, uint32_t n_windows, OrientOut* d_out) {
    if (nkp == 0) return;
    const uint32_t groups = (nkp + ORI_KPB - 1) / ORI_KPB;
    hipLaunchKernelGGL(k_orientation, dim3((groups + 7u) / 8u * 8u), dim3(ORI_NT), 0, s, lt, d_kp, nkp, (const unsigned*)nullptr,
                       window_mask, n_windows, d_out, 1u);
}
void orientation_counted(hipStream_t s, const LevelTable& lt, const KpParam* d_kp, const uint32_t* d_nkp, uint32_t max_kp,
                         unsigned long long window_mask, uint32_t n_windows, OrientOut* d_out, uint32_t out_stride) {
    if (max_kp == 0) return;
    const uint32_t groups = (max_kp + ORI_KPB - 1) / ORI_KPB;
    hipLaunchKernelGGL(k_orientation, dim3((groups + 7u) / 8u * 8u), dim3(ORI_NT), 0, s, lt, d_kp, max_kp, d_nkp, window_mask, n_windows,
                       d_out, out_stride);
}

void mldb(hipStream_t s, const LevelTable& lt, const KpParam* d_kp, const float* d_cosi, uint32_t nkp,
          uint32_t channels, uint8_t* d_desc64) {
    if (nkp == 0) return;
    const uint32_t groups = (nkp + MLDB_KPB - 1) / MLDB_KPB;
    hipLaunchKernelGGL(k_mldb, dim3((groups + 7u) / 8u * 8u), dim3(64 * MLDB_KPB), 0, s, lt, d_kp,
                       reinterpret_cast<const float2*>(d_cosi), nkp, (const unsigned*)nullptr, DeviceAngles{nullptr, 0u, 0u, nullptr}, channels,
                       d_desc64, 0u, MldbHostMirror{nullptr, nullptr, nullptr, 0u, nullptr});
}
void mldb_counted(hipStream_t s, const LevelTable& lt, const KpParam* d_kp, const uint32_t* d_nkp, uint32_t first, uint32_t last, OrientOut* d_sums,
                  uint32_t sums_stride, bool libm_fma, uint32_t* d_flag, uint32_t channels, uint8_t* d_desc64, const MldbMirror* mirror) {
    first = first / MLDB_KPB * MLDB_KPB;
    MldbHostMirror hm{nullptr, nullptr, nullptr, 0u, nullptr};
    if (mirror && sums_stride == 2)
        hm = MldbHostMirror{(uint4*)mirror->host_recs, (const uint4*)mirror->d_hdr, (uint4*)mirror->host_hdr, mirror->hdr_bytes / 16u, mirror->host_desc};
    if (last <= first && !hm.recs) return;
    const uint32_t groups = last > first ? (last - first + MLDB_KPB - 1) / MLDB_KPB : 1u;  // (a mirror's headers travel with workgroup 0 at least)
    hipLaunchKernelGGL(k_mldb, dim3((groups + 7u) / 8u * 8u), dim3(64 * MLDB_KPB), 0, s, lt, d_kp, (const float2*)nullptr, last, d_nkp,
                       DeviceAngles{d_sums, sums_stride, libm_fma ? 1u : 0u, d_flag}, channels, d_desc64, first, hm);
}
// test / self-test hook: out[i] = {atan2f(a[i], b[i]), cosf(a[i]), sinf(a[i])} as the device forms them (akz_libm.hpp)
void libm_eval(hipStream_t s, const float* a, const float* b, float* out3, uint64_t n, bool fma, uint32_t* d_flag) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_libm_eval, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a, b, out3, (size_t)n, fma ? 1u : 0u, d_flag);
}
uint32_t match_num_chunks(uint32_t n0, uint32_t n1) {
    const uint32_t qblocks = (n0 + MT - 1) / MT, tiles = std::max<uint32_t>(1, (n1 + MT - 1) / MT);
    // enough workgroups to fill 256 CUs several times over, at least 4 LDS tiles per chunk
    const uint32_t want = std::max<uint32_t>(1, 2048u / std::max<uint32_t>(1, qblocks));
    return std::max<uint32_t>(1, std::min<uint32_t>({want, (tiles + 3) / 4, 64u}));
}
// d_part: chunks * n0 records of scratch; d_out: n0 merged records
// records of every query over `chunks` chunks of the train set: d_rec[chunk * n0 + query] (merged by match_compact)
void match(hipStream_t s, const uint8_t* d0, uint32_t n0, const uint8_t* d1, uint32_t n1, uint32_t threshold,
           bool rows_le_61, uint32_t chunks, MatchRec* d_rec) {
    if (n0 == 0) return;
    const uint32_t tiles = std::max<uint32_t>(1, (n1 + MT - 1) / MT);
    const uint32_t chunk_rows = ((tiles + chunks - 1) / chunks) * MT;
    hipLaunchKernelGGL(k_match, dim3((n0 + MT - 1) / MT, chunks), dim3(MT), 0, s, reinterpret_cast<const uint4*>(d0), n0,
                       reinterpret_cast<const uint4*>(d1), n1, chunk_rows, threshold, rows_le_61 ? 0xffu : 0xffffffffu, d_rec);
}
size_t match_merge_compact_state_bytes(uint32_t n0) { return ((size_t)(n0 + 255) / 256 + 1) * sizeof(unsigned long long); }
void match_merge_compact(hipStream_t s, const MatchRec* d_part, uint32_t n0, uint32_t chunks, uint32_t threshold, double ratio2,
                         akz_match* d_out, unsigned long long* d_n_out, void* d_state, uint32_t epoch) {
    hipLaunchKernelGGL(k_match_merge_compact, dim3((n0 + 255) / 256), dim3(256), 0, s, d_part, n0, chunks, threshold, ratio2, d_out,
                       d_n_out, (unsigned long long*)d_state, epoch);
}
void match_merge(hipStream_t s, const MatchRec* d_part, uint32_t n0, uint32_t chunks, uint32_t threshold, MatchRec* d_out) {
    hipLaunchKernelGGL(k_match_merge, dim3((n0 + 255) / 256), dim3(256), 0, s, d_part, n0, chunks, threshold, d_out);
}
void match_compact(hipStream_t s, const MatchRec* d_rec, uint32_t n0, uint32_t chunks, uint32_t threshold, double ratio2,
                   akz_match* d_out, unsigned long long* d_n_out) {
    hipLaunchKernelGGL(k_match_compact, dim3(1), dim3(1024), 0, s, d_rec, n0, chunks, threshold, ratio2, d_out, d_n_out);
}
void match_compact_sets(hipStream_t s, const MatchRec* d_rec, uint32_t n0, uint32_t n_sets, uint32_t chunks, uint32_t threshold,
                        double ratio2, akz_match* d_out, unsigned long long* d_n_out) {
    hipLaunchKernelGGL(k_match_compact, dim3(n_sets), dim3(1024), 0, s, d_rec, n0, chunks, threshold, ratio2, d_out, d_n_out);
}

}  // namespace launch
}  // namespace akz
