// k_octave_resident (gfx950): the coarse end of the pyramid as ONE launch per batch, one workgroup per image.
//
// From the level where a whole image fits a compute unit (at most 512 patches of 8 x 8 pixels: 240 x 135, the fourth
// octave of a 1080p frame) the reference's per-level sequence (lib.rs:78-119)
//
//     Lt_i = half_size(Lt_{i-1}) | clone      image.rs:102-118, lib.rs:82 / :92
//     Lsmooth_i = gaussian_blur(Lt_i, 1.0)    image.rs:374-380 (H then V pass, fill_border after each, :239-332)
//     Lx, Ly = scharr(Lsmooth_i, sigma 1)     derivatives.rs:41-130
//     Lflow_i = pm_g2(Lx, Ly, k)              lib.rs:26-41 (f64 inside)
//     n_i x calculate_step(Lt_i, Lflow_i)     nonlinear_diffusion.rs:15-173
//
// runs for ALL remaining levels (and octaves) without leaving the chip: a 512-thread workgroup (two waves per SIMD,
// 256 VGPRs each) keeps the image in registers -- every thread owns an 8 x 8 patch of Lt and of Lflow for the whole
// launch -- and touches HBM only to read the previous level's Lt once and to write the planes the result keeps
// (Lsmooth, Lflow, the final Lt and, if wanted, Lstep of every level).  As separate launches these levels are chains
// of 17 (1080p, 4 x 4) to 30 dependent dispatches of a few hundred workgroups each, bound by dispatch-to-dispatch
// latency and, in a batch, competing for compute units with the bandwidth-bound kernels of the fine octaves.
//
// Diffusion step: every thread publishes the four edges of its patch (8 values each) in LDS, one barrier, reads its
// neighbours' edges, updates its 64 pixels in registers.  Rows r and r+4 of a patch share a register pair, so the
// update is packed f32 arithmetic throughout.  Image borders cost nothing per step: nonlinear_diffusion.rs:84-137
// drops the flux towards a missing neighbour, which equals the interior expression ((xpos - xneg) + ypos) - yneg with
// that flux set to zero (x - 0 = x, 0 - x = -x, and the last row's `ypos` towards y-1 is -yneg exactly); a flux is
// (c + c_nb) * (L_nb - L), and the Lflow value of a pixel beyond the border is defined as the NEGATIVE of the border
// pixel's, which makes that pair sum an exact zero.  Only the sign of a zero result can differ.
//
// Level preparation: the separable passes go through one LDS plane of the image exactly as the reference runs them
// (H pass at the pixel's own column, fill_border = reading the plane at clamped coordinates in the V pass).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "akz_internal.hpp"
#include "akz_pm_g2.hpp"

namespace akz {
namespace {

typedef float v2 __attribute__((ext_vector_type(2)));

constexpr int RNT = 512;                    // threads per workgroup: 8 waves, two per SIMD
constexpr int kResLevels = launch::kResidentMaxLevels;
constexpr int kResSteps = launch::kResidentMaxSteps;
constexpr int kPlaneFloats = RNT * 64;      // one image plane (padded to whole patches) == the four edge arrays

struct ResLevel {
    float *lt, *lsmooth, *lflow, *lstep;  // image 0 of the batch; images w*h apart.  lstep may be null
    int w, h;
    int half;        // the level opens an octave: its input is the 2x2 mean of the current image
    int n_tau, tau0; // diffusion steps and the index of the first one's 0.5f * tau
    unsigned k_pow;  // octave: the contrast factor is k * 0.75^k_pow (lib.rs:84)
};
struct ResArgs {
    const float* prev;  // final Lt of the level before the first resident one, image 0; images pw*ph apart
    int pw, ph;
    int n_levels;
    float g0, g1, g2;   // gaussian_kernel(1.0, 3)
    float kn, kwn;      // scale-1 Scharr main-axis taps [kn, kwn, kn]
    const double* d_k;  // contrast factor per image
    ResLevel lv[kResLevels];
    float half_tau[kResSteps];
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ double octave_contrast(double k, unsigned pow) {  // lib.rs:84, one octave at a time
    for (unsigned i = 0; i < pow; ++i) k = k * 0.75;
    return k;
}

// Patch grid of one level.  Thread t owns the patch (bx, by) = (t % XG, t / XG): pixels x0 .. x0+7, y0 .. y0+7.
struct Geo {
    int w, h, XG, YG, PITCH;
    int bx, by, x0, y0;
    bool active;
};
__device__ __forceinline__ Geo make_geo(int w, int h, int tid) {
    Geo g;
    g.w = w; g.h = h;
    g.XG = (w + 7) >> 3; g.YG = (h + 7) >> 3; g.PITCH = g.XG << 3;
    g.by = tid / g.XG; g.bx = tid - g.by * g.XG;
    g.active = g.by < g.YG;
    if (!g.active) g.bx = g.by = 0;  // threads beyond the patch grid shadow patch (0, 0) and store nothing: every
                                     // thread runs the same code, so register patches die where the code says they do
    g.x0 = g.bx << 3; g.y0 = g.by << 3;
    return g;
}
// LDS plane: pixel (x, y) at y * PITCH + (x % 8) * XG + x / 8 -- for a fixed column of the patch the threads of a wave
// (consecutive bx) read and write consecutive words.
__device__ __forceinline__ int pcol(const Geo& g, int x) { return (x & 7) * g.XG + (x >> 3); }

// register patch: R[k][i] = {pixel (x0+i, y0+k), pixel (x0+i, y0+k+4)}
#define AKZ_FOR_PATCH(k, c, i) \
    _Pragma("unroll") for (int k = 0; k < 4; ++k) _Pragma("unroll") for (int c = 0; c < 2; ++c) _Pragma("unroll") for (int i = 0; i < 8; ++i)

__device__ __forceinline__ void plane_put(float* __restrict__ sm, const Geo& g, const v2 (&R)[4][8]) {
    if (!g.active) return;
    int base = g.y0 * g.PITCH + g.bx;
    asm volatile("" : "+v"(base));  // as in plane_vpass: addresses are not carried from pass to pass
    AKZ_FOR_PATCH(k, c, i) sm[base + (k + 4 * c) * g.PITCH + i * g.XG] = R[k][i][c];
}
// the patch columns' left and right neighbours in the plane (0 beyond the patch grid: only results that are never
// read depend on them)
__device__ __forceinline__ void plane_sides(const float* __restrict__ sm, const Geo& g, v2 (&L)[4], v2 (&Rt)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            int row = (g.y0 + k + 4 * c) * g.PITCH;
            asm volatile("" : "+v"(row));
            L[k][c] = g.bx > 0 ? sm[row + 7 * g.XG + g.bx - 1] : 0.0f;
            Rt[k][c] = g.bx + 1 < g.XG ? sm[row + g.bx + 1] : 0.0f;
        }
}
// V pass of a 3-tap filter read from the plane with fill_border (image.rs:239-260, :300-332): the value at (x, y) is the
// interior value at (clamp(x,1,w-2), clamp(y,1,h-2)), whose taps are rows of the border-filled H pass, i.e. plane
// rows clamped once more.  DIFF: taps [-1, 0, 1] evaluated as (0 - a) + c.
template <bool DIFF>
__device__ __forceinline__ void plane_vpass(const float* __restrict__ sm, const Geo& g, float t0, float t1, float t2,
                                            v2 (&O)[4][8]) {
    int col[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        col[i] = pcol(g, clampi(g.x0 + i, 1, g.w - 2));
        // opaque to the optimiser: LDS addresses are formed anew in every pass (one add each) instead of being kept --
        // i.e. spilled -- across the passes of a level
        asm volatile("" : "+v"(col[i]));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // rows y0 + k (.x) and y0 + k + 4 (.y)
        int ra[2], rb[2], rc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int cy = clampi(g.y0 + k + 4 * c, 1, g.h - 2);
            ra[c] = clampi(cy - 1, 1, g.h - 2) * g.PITCH;
            rb[c] = cy * g.PITCH;
            rc[c] = clampi(cy + 1, 1, g.h - 2) * g.PITCH;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const v2 a = v2{sm[ra[0] + col[i]], sm[ra[1] + col[i]]}, cc = v2{sm[rc[0] + col[i]], sm[rc[1] + col[i]]};
            if (DIFF) {
                O[k][i] = (0.0f - a) + cc;
            } else {
                const v2 b = v2{sm[rb[0] + col[i]], sm[rb[1] + col[i]]};
                O[k][i] = ((0.0f + t0 * a) + t1 * b) + t2 * cc;
            }
            asm volatile("" : "+v"(O[k][i]));  // evaluated HERE: sunk towards its use, the result would leave its 2-3 operands alive instead
        }
        __builtin_amdgcn_sched_barrier(0);  // one pair of rows in flight, not all four: the patch arrays own the registers
    }
}
// ---- global memory: raw buffer accesses (as in akz_march.hip) -------------------------------------------------------
// rsrc = one image's plane, num_records = its bytes: the hardware drops a store and returns 0 for a load whose offset
// is out of range, so pixels of a patch that lie beyond the image are handled by presenting VO_NONE -- straight-line
// code, no exec-mask branch per row (with branches the register allocator spilled around every one of them).
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr unsigned VO_NONE = 0x40000000u;
constexpr int RSRC_FLAGS = 0x00020000;  // raw buffer, dword data format (gfx9 word 3)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const float* plane_of_image, int w, int h) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane_of_image), 0, w * h * 4, RSRC_FLAGS);
}
// one image plane's pixels of the patch to global memory
// VEC (a property of the launch): every plane width is a multiple of 4, i.e. rows start 16-byte aligned and end on a group of four
template <bool VEC>
__device__ __forceinline__ void patch_store(float* __restrict__ dst, const Geo& g, const v2 (&R)[4][8]) {
    const __amdgpu_buffer_rsrc_t rs = plane_rsrc(dst, g.w, g.h);
    constexpr bool vec = VEC;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int y = g.y0 + k + 4 * c;
            const bool row = g.active && y < g.h;
            const unsigned o = (unsigned)(y * g.w + g.x0) * 4u;
            if (vec) {
                const u4 lo = {__float_as_uint(R[k][0][c]), __float_as_uint(R[k][1][c]), __float_as_uint(R[k][2][c]), __float_as_uint(R[k][3][c])};
                const u4 hi = {__float_as_uint(R[k][4][c]), __float_as_uint(R[k][5][c]), __float_as_uint(R[k][6][c]), __float_as_uint(R[k][7][c])};
                __builtin_amdgcn_raw_buffer_store_b128(lo, rs, (int)(row && g.x0 + 4 <= g.w ? o : VO_NONE), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(hi, rs, (int)(row && g.x0 + 8 <= g.w ? o + 16u : VO_NONE), 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(R[k][i][c]), rs, (int)(row && g.x0 + i < g.w ? o + 4u * i : VO_NONE), 0, 0);
            }
        }
}
// ... and back (pixels beyond the image read as 0)
template <bool VEC>
__device__ __forceinline__ void patch_load(const float* __restrict__ src, const Geo& g, v2 (&R)[4][8]) {
    const __amdgpu_buffer_rsrc_t rs = plane_rsrc(src, g.w, g.h);
    constexpr bool vec = VEC;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int y = g.y0 + k + 4 * c;
            const bool row = y < g.h;
            const unsigned o = (unsigned)(y * g.w + g.x0) * 4u;
            if (vec) {
                const u4 lo = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(row && g.x0 + 4 <= g.w ? o : VO_NONE), 0, 0);
                const u4 hi = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(row && g.x0 + 8 <= g.w ? o + 16u : VO_NONE), 0, 0);
                R[k][0][c] = __uint_as_float(lo.x); R[k][1][c] = __uint_as_float(lo.y); R[k][2][c] = __uint_as_float(lo.z); R[k][3][c] = __uint_as_float(lo.w);
                R[k][4][c] = __uint_as_float(hi.x); R[k][5][c] = __uint_as_float(hi.y); R[k][6][c] = __uint_as_float(hi.z); R[k][7][c] = __uint_as_float(hi.w);
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    R[k][i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(row && g.x0 + i < g.w ? o + 4u * i : VO_NONE), 0, 0));
            }
        }
}
// the 2x2 mean of image.rs:102-118 of a pw x ph plane (or, !HALF, the plane itself) into the patch; 0 beyond the image
template <bool VEC>
__device__ __forceinline__ void patch_input(const float* __restrict__ src, int pw, int ph, bool half, const Geo& g, v2 (&R)[4][8]) {
    if (!half) {
        patch_load<VEC>(src, g, R);
        return;
    }
    const __amdgpu_buffer_rsrc_t rs = plane_rsrc(src, pw, ph);
    constexpr bool vec = VEC;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int y = g.y0 + k + 4 * c;
            const bool row = y < g.h;
            const unsigned o = (unsigned)(2 * y * pw + 2 * g.x0) * 4u;  // input row 2y, column 2 x0
            float t[16], b[16];                                          // input rows 2y and 2y+1, columns 2 x0 .. 2 x0 + 15
            if (vec) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool in = row && 2 * g.x0 + 4 * q < pw;
                    const u4 u = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(in ? o + 16u * q : VO_NONE), 0, 0);
                    const u4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(in ? o + 16u * q + 4u * pw : VO_NONE), 0, 0);
                    t[4 * q] = __uint_as_float(u.x); t[4 * q + 1] = __uint_as_float(u.y); t[4 * q + 2] = __uint_as_float(u.z); t[4 * q + 3] = __uint_as_float(u.w);
                    b[4 * q] = __uint_as_float(d.x); b[4 * q + 1] = __uint_as_float(d.y); b[4 * q + 2] = __uint_as_float(d.z); b[4 * q + 3] = __uint_as_float(d.w);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const bool in = row && 2 * g.x0 + q < pw;
                    t[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(in ? o + 4u * q : VO_NONE), 0, 0));
                    b[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(in ? o + 4u * q + 4u * pw : VO_NONE), 0, 0));
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = 0.0f;
                v = v + t[2 * i];
                v = v + b[2 * i];
                v = v + t[2 * i + 1];
                v = v + b[2 * i + 1];
                v = v / 4.0f;
                R[k][i][c] = (row && g.x0 + i < g.w) ? v : 0.0f;
                asm volatile("" : "+v"(R[k][i][c]));
            }
        }
}

__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void st8(float* p, float a, float b, float c, float d, float e, float f, float g, float h) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
    *reinterpret_cast<float4*>(p + 4) = make_float4(e, f, g, h);
}
__device__ __forceinline__ void ld4v(const float* p, v2 (&v)[4]) {  // a column edge: rows 0,4 | 1,5 | 2,6 | 3,7
    float t[8];
    ld8(p, t);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = v2{t[2 * k], t[2 * k + 1]};
}

// Edge arrays in LDS (they alias the preparation plane): [parity][thread][8] each
struct Edges {
    float *top, *bot, *left, *right;
};
__device__ __forceinline__ Edges edges_of(float* sm, int parity) {
    float* b = sm + parity * (RNT * 8);
    return Edges{b, b + 2 * RNT * 8, b + 4 * RNT * 8, b + 6 * RNT * 8};
}
__device__ __forceinline__ void edges_put(const Edges& e, int tid, const v2 (&R)[4][8]) {
    st8(e.top + tid * 8, R[0][0].x, R[0][1].x, R[0][2].x, R[0][3].x, R[0][4].x, R[0][5].x, R[0][6].x, R[0][7].x);
    st8(e.bot + tid * 8, R[3][0].y, R[3][1].y, R[3][2].y, R[3][3].y, R[3][4].y, R[3][5].y, R[3][6].y, R[3][7].y);
    st8(e.left + tid * 8, R[0][0].x, R[0][0].y, R[1][0].x, R[1][0].y, R[2][0].x, R[2][0].y, R[3][0].x, R[3][0].y);
    st8(e.right + tid * 8, R[0][7].x, R[0][7].y, R[1][7].x, R[1][7].y, R[2][7].x, R[2][7].y, R[3][7].x, R[3][7].y);
}
struct Nbr {
    int up, down, left, right;  // thread ids (the thread's own where there is no neighbour patch)
};

// One explicit diffusion step (nonlinear_diffusion.rs:30-143) on the patch.  A flux between two pixels is evaluated once,
// as (c + c') * (L' - L) with ' the pixel to the right / below: it is `xpos` / `ypos` of the one and `xneg` / `yneg` of
// the other (:63-66).  Rows are walked top to bottom with the flux above carried along; the flux between rows 3 and 4
// (the two halves of the register pairs) is formed first.
template <bool STEP_OUT, bool VEC>
__device__ __forceinline__ void fed_step(v2 (&P)[4][8], const v2 (&C)[4][8], const float (&CT)[8], const float (&CB)[8],
                                         const v2 (&CL)[4], const v2 (&CR)[4], const Edges& rd, const Nbr& nb,
                                         const float ht, float* __restrict__ lstep, const Geo& g) {
    // live registers are the budget here (the two patches and the Lflow edges are 160 of 256): edges are loaded where
    // they are first needed and the scheduler is kept from interleaving the rows
    v2 LL[4], LR[4];
    float seam[8];  // flux between rows 3 and 4
    v2 yprev[8];
    {
        float LT[8];
        ld8(rd.bot + nb.up * 8, LT);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            seam[i] = (C[3][i].x + C[0][i].y) * (P[0][i].y - P[3][i].x);
            yprev[i] = v2{(CT[i] + C[0][i].x) * (P[0][i].x - LT[i]), seam[i]};
        }
    }
    ld4v(rd.right + nb.left * 8, LL);
    ld4v(rd.left + nb.right * 8, LR);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float LB[8];
        if (k == 3) ld8(rd.top + nb.down * 8, LB);
        v2 xprev = (CL[k] + C[k][0]) * (P[k][0] - LL[k]);
        v2 ST[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v2 ynext;
            if (k < 3) ynext = (C[k][i] + C[k < 3 ? k + 1 : 3][i]) * (P[k < 3 ? k + 1 : 3][i] - P[k][i]);
            else ynext = v2{seam[i], (C[3][i].y + CB[i]) * (LB[i] - P[3][i].y)};
            const v2 cn = i < 7 ? C[k][i < 7 ? i + 1 : 7] : CR[k];
            const v2 pn = i < 7 ? P[k][i < 7 ? i + 1 : 7] : LR[k];
            const v2 xnext = (C[k][i] + cn) * (pn - P[k][i]);
            const v2 st = ht * (((xnext - xprev) + ynext) - yprev[i]);
            P[k][i] = P[k][i] + st;  // the old value is in no later flux: xnext / ynext carry it on
            xprev = xnext;
            yprev[i] = ynext;
            if (STEP_OUT) ST[i] = st;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (STEP_OUT) {
            const __amdgpu_buffer_rsrc_t rs = plane_rsrc(lstep, g.w, g.h);
            constexpr bool vec = VEC;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int y = g.y0 + k + 4 * c;
                const bool row = g.active && y < g.h;
                const unsigned o = (unsigned)(y * g.w + g.x0) * 4u;
                if (vec) {
                    const u4 lo = {__float_as_uint(ST[0][c]), __float_as_uint(ST[1][c]), __float_as_uint(ST[2][c]), __float_as_uint(ST[3][c])};
                    const u4 hi = {__float_as_uint(ST[4][c]), __float_as_uint(ST[5][c]), __float_as_uint(ST[6][c]), __float_as_uint(ST[7][c])};
                    __builtin_amdgcn_raw_buffer_store_b128(lo, rs, (int)(row && g.x0 + 4 <= g.w ? o : VO_NONE), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(hi, rs, (int)(row && g.x0 + 8 <= g.w ? o + 16u : VO_NONE), 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ST[i][c]), rs, (int)(row && g.x0 + i < g.w ? o + 4u * i : VO_NONE), 0, 0);
                }
            }
        }
    }
}

template <bool VEC>
__global__ void __launch_bounds__(RNT) __attribute__((amdgpu_waves_per_eu(2, 2)))
k_octave_resident(const ResArgs a) {
    __shared__ __attribute__((aligned(16))) float sm[kPlaneFloats];
    const int tid = threadIdx.x;
    const size_t img = blockIdx.x;
    v2 P[4][8];  // Lt
    v2 C[4][8];  // Lflow (negated copies beyond the image border)

    // ---- the first level's input from global memory: clone (lib.rs:92) or 2x2 mean (image.rs:102-118) ----
    Geo g = make_geo(a.lv[0].w, a.lv[0].h, tid);
    patch_input<VEC>(a.prev + img * (size_t)a.pw * (size_t)a.ph, a.pw, a.ph, a.lv[0].half != 0, g, P);

    for (int l = 0; l < a.n_levels; ++l) {
        const ResLevel& lv = a.lv[l];
        if (l > 0 && lv.half) {  // a new octave from the resident image: through the plane, then a coarser patch grid
            plane_put(sm, g, P);
            __syncthreads();
            const Geo o = g;
            g = make_geo(lv.w, lv.h, tid);
            AKZ_FOR_PATCH(k, c, i) {
                const int x = g.x0 + i, y = g.y0 + k + 4 * c;
                float v = 0.0f;
                if (x < g.w && y < g.h) {
                    const int r0 = (2 * y) * o.PITCH, r1 = r0 + o.PITCH, c0 = pcol(o, 2 * x), c1 = pcol(o, 2 * x + 1);
                    v = v + sm[r0 + c0];
                    v = v + sm[r1 + c0];
                    v = v + sm[r0 + c1];
                    v = v + sm[r1 + c1];
                    v = v / 4.0f;
                }
                P[k][i][c] = v;
            }
            __syncthreads();
        }
        const size_t off = img * (size_t)g.w * (size_t)g.h;
        const double kc = octave_contrast(a.d_k[img], lv.k_pow);
        const double inverse_k = 1.0 / (kc * kc);

        // ---- Lsmooth = gaussian_blur(Lt, 1.0): H pass at the own columns, V pass from the border-filled plane ----
        plane_put(sm, g, P);
        __syncthreads();
        {
            v2 T[4][8];
            v2 sl[4], sr[4];
            plane_sides(sm, g, sl, sr);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const v2 lft = i > 0 ? P[k][i > 0 ? i - 1 : 0] : sl[k], rgt = i < 7 ? P[k][i < 7 ? i + 1 : 7] : sr[k];
                    T[k][i] = ((0.0f + a.g0 * lft) + a.g1 * P[k][i]) + a.g2 * rgt;
                    asm volatile("" : "+v"(T[k][i]));  // (evaluated here, see plane_vpass)
                }
            // Lt is not needed again before the diffusion steps: it waits in the level's Lt plane (16 wide stores and
            // loads per thread) instead of occupying a quarter of the register file during the preparation
            patch_store<VEC>(lv.lt + off, g, P);
            __syncthreads();
            plane_put(sm, g, T);
            __syncthreads();
            plane_vpass<false>(sm, g, a.g0, a.g1, a.g2, T);  // T = Lsmooth
            patch_store<VEC>(lv.lsmooth + off, g, T);
            __syncthreads();
            plane_put(sm, g, T);
            __syncthreads();
            // ---- scale-1 Scharr pair (derivatives.rs:41-65): "Lx" = V_diff(H_main), "Ly" = V_main(H_diff); pm_g2 ----
            // (one plane, three live register patches)
            plane_sides(sm, g, sl, sr);
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // both H passes at once: H_main into C, H_diff in place of Lsmooth
                v2 d[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const v2 lft = i > 0 ? T[k][i > 0 ? i - 1 : 0] : sl[k], rgt = i < 7 ? T[k][i < 7 ? i + 1 : 7] : sr[k];
                    C[k][i] = ((0.0f + a.kn * lft) + a.kwn * T[k][i]) + a.kn * rgt;
                    d[i] = (0.0f - lft) + rgt;
                    asm volatile("" : "+v"(C[k][i]), "+v"(d[i]));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) T[k][i] = d[i];
            }
            __syncthreads();
            plane_put(sm, g, C);
            __syncthreads();
            plane_vpass<true>(sm, g, 0.0f, 0.0f, 0.0f, C);  // C = Lx
            __syncthreads();
            plane_put(sm, g, T);
            __syncthreads();
            plane_vpass<false>(sm, g, a.kn, a.kwn, a.kn, T);  // T = Ly
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    C[k][i].x = pm_g2_px(C[k][i].x, T[k][i].x, inverse_k);
                    C[k][i].y = pm_g2_px(C[k][i].y, T[k][i].y, inverse_k);
                    asm volatile("" : "+v"(C[k][i]));
                    if (i & 1) __builtin_amdgcn_sched_barrier(0);  // four divisions in flight, not sixty-four
                }
            patch_store<VEC>(lv.lflow + off, g, C);
            // beyond the image: the negative of the border pixel's value (rows after columns)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float edge = 0.0f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (g.x0 + i < g.w) edge = C[k][i][c];
                        else C[k][i][c] = -edge;
                    }
                }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float edge = 0.0f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (g.y0 + r < g.h) edge = C[r & 3][i][r >> 2];
                    else C[r & 3][i][r >> 2] = -edge;
                }
            }
        }
        __syncthreads();  // the plane is dead: its memory becomes the edge arrays

        // ---- edges of Lflow (once per level), then the diffusion steps ----
        Nbr nb;
        nb.up = g.active && g.by > 0 ? tid - g.XG : tid;
        nb.down = g.active && g.by + 1 < g.YG ? tid + g.XG : tid;
        nb.left = g.active && g.bx > 0 ? tid - 1 : tid;
        nb.right = g.active && g.bx + 1 < g.XG ? tid + 1 : tid;
        float CT[8], CB[8];
        v2 CL[4], CR[4];
        {
            const Edges e0 = edges_of(sm, 0), e1 = edges_of(sm, 1);
            edges_put(e0, tid, C);
            __syncthreads();
            ld8(e0.bot + nb.up * 8, CT);
            ld8(e0.top + nb.down * 8, CB);
            ld4v(e0.right + nb.left * 8, CL);
            ld4v(e0.left + nb.right * 8, CR);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (g.by == 0) CT[i] = -C[0][i].x;
                if (g.by + 1 >= g.YG) CB[i] = -C[3][i].y;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (g.bx == 0) CL[k] = -C[k][0];
                if (g.bx + 1 >= g.XG) CR[k] = -C[k][7];
            }
            patch_load<VEC>(lv.lt + off, g, P);
            edges_put(e1, tid, P);
            __syncthreads();
        }
        int par = 1;
        const float* ht = a.half_tau + lv.tau0;
        for (int s = 0; s + 1 < lv.n_tau; ++s) {
            const Edges rd = edges_of(sm, par), wr = edges_of(sm, par ^ 1);
            fed_step<false, VEC>(P, C, CT, CB, CL, CR, rd, nb, ht[s], nullptr, g);
            edges_put(wr, tid, P);
            __syncthreads();
            par ^= 1;
        }
        {
            const Edges rd = edges_of(sm, par);
            if (lv.lstep) fed_step<true, VEC>(P, C, CT, CB, CL, CR, rd, nb, ht[lv.n_tau - 1], lv.lstep + off, g);
            else fed_step<false, VEC>(P, C, CT, CB, CL, CR, rd, nb, ht[lv.n_tau - 1], nullptr, g);
            patch_store<VEC>(lv.lt + off, g, P);
        }
        __syncthreads();  // the edge arrays are dead: the next level's plane overwrites them
    }
}

}  // namespace

namespace launch {

bool octave_resident_supported(uint32_t w, uint32_t h) {
    return w >= 8 && h >= 8 && (uint64_t)((w + 7) / 8) * ((h + 7) / 8) <= (uint64_t)RNT;
}

void octave_resident(hipStream_t s, const float* prev, uint32_t pw, uint32_t ph, uint32_t n, const ResidentLevel* levels,
                     uint32_t n_levels, const float* g3, const double* d_k) {
    ResArgs a;
    a.prev = prev;
    a.pw = (int)pw; a.ph = (int)ph;
    a.n_levels = (int)n_levels;
    a.g0 = g3[0]; a.g1 = g3[1]; a.g2 = g3[2];
    std::vector<float> m, o;
    scharr_kernels(1, m, o);
    a.kn = m[0]; a.kwn = m[1];
    a.d_k = d_k;
    int t = 0;
    for (uint32_t l = 0; l < n_levels; ++l) {
        const ResidentLevel& L = levels[l];
        ResLevel& d = a.lv[l];
        d.lt = L.lt; d.lsmooth = L.lsmooth; d.lflow = L.lflow; d.lstep = L.lstep;
        d.w = (int)L.w; d.h = (int)L.h;
        d.half = L.half ? 1 : 0;
        d.n_tau = (int)L.n_tau; d.tau0 = t;
        d.k_pow = L.k_pow;
        for (uint32_t j = 0; j < L.n_tau; ++j) a.half_tau[t++] = L.half_tau[j];
    }
    for (uint32_t l = n_levels; l < (uint32_t)kResLevels; ++l) a.lv[l] = ResLevel{};
    for (; t < kResSteps; ++t) a.half_tau[t] = 0.0f;
    bool vec = (pw & 3) == 0;
    for (uint32_t l = 0; l < n_levels; ++l) vec = vec && (levels[l].w & 3) == 0;
    if (vec) hipLaunchKernelGGL(k_octave_resident<true>, dim3(n), dim3(RNT), 0, s, a);
    else hipLaunchKernelGGL(k_octave_resident<false>, dim3(n), dim3(RNT), 0, s, a);
}

}  // namespace launch
}  // namespace akz
