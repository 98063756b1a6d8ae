// Fused separable-stencil chains for gfx950, staged through LDS.
//
// The reference builds every derived plane from full-image separable passes, each followed by
// fill_border (akaze/src/types/image.rs:239-332): after a pass with half width hw the value at
// (x, y) is the valid-interior result at (clamp(x, hw, w-1-hw), clamp(y, hw, h-1-hw)).  A chain of
// such passes is evaluated here per 64x32 output tile entirely in LDS:
//
//   * each stage keeps a rectangular window of its plane in LDS; window k-1 is window k grown by
//     the half width of pass k in the pass direction;
//   * every window position holds the FILLED value of that stage, i.e. the raw result at its
//     clamped coordinates, so the next stage reads its taps at (cx + off, cy) or (cx, cy + off)
//     without further clamping;
//   * tiles are always full size: the last tile of a row/column is shifted inward
//     (tile_origin) and recomputes a few pixels of its neighbour with identical results.  With
//     full tiles every clamped read stays inside the previous window.
//
// Arithmetic is the reference's: f32 mul then add, taps left to right starting from 0.0f, no FMA.
#include <hip/hip_runtime.h>

#include "akz_internal.hpp"

namespace akz {
namespace {

#ifndef AKZ_STENCIL_NT
#define AKZ_STENCIL_NT 512
#endif
constexpr int TW = 64, TH = 32, NT = AKZ_STENCIL_NT;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// origin of tile b along a dimension of `dim` pixels
__device__ __forceinline__ int tile_origin(int b, int tile, int dim) {
    const int o = b * tile;
    return (dim > tile && o + tile > dim) ? dim - tile : o;
}

__device__ __forceinline__ float unit_px(const float* p, size_t i) { return p[i]; }
// create_unit_float_image: f32::from(v) * 1f32 / 255f32 (types/image.rs:136)
__device__ __forceinline__ float unit_px(const uint8_t* p, size_t i) { return ((float)p[i] * 1.0f) / 255.0f; }

__device__ __forceinline__ double octave_contrast(double k, unsigned pow) {
    for (unsigned i = 0; i < pow; ++i) k = k * 0.75;  // lib.rs:84, one octave at a time in f64
    return k;
}
__device__ __forceinline__ float pm_g2_px(float lx, float ly, double inverse_k) {  // lib.rs:30-37
    const double dx = (double)lx, dy = (double)ly;
    return (float)(1.0 / (1.0 + inverse_k * (dx * dx + dy * dy)));
}

struct DenseTaps {
    float k[kMaxTaps];
};

// ---------------------------------------------------------------------------------------------
// gaussian_blur = V(H(in)) with a dense (2*HW+1)-tap kernel (types/image.rs:374-380)
// ---------------------------------------------------------------------------------------------
template <int HW, typename T>
__global__ void __launch_bounds__(NT)
k_blur(const T* __restrict__ in, float* __restrict__ out, int w, int h, DenseTaps t) {
    constexpr int RW = TW + 2 * HW, RH = TH + 2 * HW;
    __shared__ float sIn[RH * RW];
    __shared__ float sH[RH * TW];
    const int tid = threadIdx.x;
    const int x0 = tile_origin(blockIdx.x, TW, w), y0 = tile_origin(blockIdx.y, TH, h);
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    for (int idx = tid; idx < RH * RW; idx += NT) {
        const int ly = idx / RW, lx = idx - ly * RW;
        const int gx = x0 - HW + lx, gy = y0 - HW + ly;
        sIn[idx] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? unit_px(in, base + (size_t)gy * w + gx) : 0.0f;
    }
    __syncthreads();
    for (int idx = tid; idx < RH * TW; idx += NT) {  // H pass on window rows y0-HW .. y0+TH+HW
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0 + lx, y = y0 - HW + ly;
        if (x < w && y >= 0 && y < h) {
            const int cx = clampi(x, HW, w - 1 - HW), cy = clampi(y, HW, h - 1 - HW);
            const float* p = sIn + (cy - (y0 - HW)) * RW + (cx - x0);  // tap 0 sits at cx - HW
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 2 * HW + 1; ++i) acc = acc + t.k[i] * p[i];
            sH[idx] = acc;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < TH * TW; idx += NT) {  // V pass on the centre
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0 + lx, y = y0 + ly;
        if (x < w && y < h) {
            const int cx = clampi(x, HW, w - 1 - HW), cy = clampi(y, HW, h - 1 - HW);
            const float* p = sH + (cy - y0) * TW + (cx - x0);  // window row of cy - HW
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 2 * HW + 1; ++i) acc = acc + t.k[i] * p[i * TW];
            out[base + (size_t)y * w + x] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// One level's preparation (lib.rs:80-105): Lt_i (clone or half_size of the previous level) ->
// Lsmooth_i = gaussian_blur(Lt_i, 1.0) -> scale-1 Scharr pair -> Lflow_i = pm_g2.  Four passes
// with hw = 1; windows: input +-2, H_g rows +-2 cols +-1, Lsmooth +-1, H_scharr rows +-1.
// With HALF the input window is computed from the previous level by the 2x2 mean of
// types/image.rs:102-118 and the centre is also written out as the level's starting Lt.
// ---------------------------------------------------------------------------------------------
template <bool HALF>
__global__ void __launch_bounds__(NT)
k_prep(const float* __restrict__ prev, float* __restrict__ lt_out, float* __restrict__ lsmooth,
       float* __restrict__ lflow, int w, int h, int pw, int ph, float g0, float g1, float g2, float kn, float kwn,
       const double* __restrict__ d_k, unsigned k_pow) {
    constexpr int IW = TW + 4, IH = TH + 4;  // input window, origin (x0-2, y0-2)
    constexpr int AW = TW + 2, AH = TH + 4;  // H_g,       origin (x0-1, y0-2)
    constexpr int BW = TW + 2, BH = TH + 2;  // Lsmooth,   origin (x0-1, y0-1)
    constexpr int CH = TH + 2;               // H_scharr,  origin (x0,   y0-1), width TW
    __shared__ float sI[IH * IW];
    __shared__ float sA[AH * AW];
    __shared__ float sB[BH * BW];
    __shared__ float sM[CH * TW];
    __shared__ float sO[CH * TW];
    const int tid = threadIdx.x;
    const int x0 = tile_origin(blockIdx.x, TW, w), y0 = tile_origin(blockIdx.y, TH, h);
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    const float* src = prev + (size_t)blockIdx.z * (size_t)pw * (size_t)ph;
    for (int idx = tid; idx < IH * IW; idx += NT) {
        const int ly = idx / IW, lx = idx - ly * IW;
        const int gx = x0 - 2 + lx, gy = y0 - 2 + ly;
        float v = 0.0f;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
            if (HALF) {
                const float* q = src + (size_t)(2 * gy) * pw + 2 * gx;
                v = v + q[0];
                v = v + q[pw];
                v = v + q[1];
                v = v + q[pw + 1];
                v = v / 4.0f;
                if (lx >= 2 && lx < TW + 2 && ly >= 2 && ly < TH + 2) lt_out[base + (size_t)gy * w + gx] = v;
            } else {
                v = src[(size_t)gy * w + gx];
            }
        }
        sI[idx] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < AH * AW; idx += NT) {  // A = H_g(in)
        const int ly = idx / AW, lx = idx - ly * AW;
        const int x = x0 - 1 + lx, y = y0 - 2 + ly;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            const int cx = clampi(x, 1, w - 2), cy = clampi(y, 1, h - 2);
            const float* p = sI + (cy - (y0 - 2)) * IW + (cx - (x0 - 2));
            sA[idx] = ((0.0f + g0 * p[-1]) + g1 * p[0]) + g2 * p[1];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < BH * BW; idx += NT) {  // B = Lsmooth = V_g(A)
        const int ly = idx / BW, lx = idx - ly * BW;
        const int x = x0 - 1 + lx, y = y0 - 1 + ly;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            const int cx = clampi(x, 1, w - 2), cy = clampi(y, 1, h - 2);
            const float* p = sA + (cy - (y0 - 2)) * AW + (cx - (x0 - 1));
            const float v = ((0.0f + g0 * p[-AW]) + g1 * p[0]) + g2 * p[AW];
            sB[idx] = v;
            if (lx >= 1 && lx <= TW && ly >= 1 && ly <= TH) lsmooth[base + (size_t)y * w + x] = v;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < CH * TW; idx += NT) {  // H passes of the Scharr pair (derivatives.rs:41-65)
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0 + lx, y = y0 - 1 + ly;
        if (x < w && y >= 0 && y < h) {
            const int cx = clampi(x, 1, w - 2), cy = clampi(y, 1, h - 2);
            const float* p = sB + (cy - (y0 - 1)) * BW + (cx - (x0 - 1));
            const float a = p[-1], b = p[0], c = p[1];
            sM[idx] = ((0.0f + kn * a) + kwn * b) + kn * c;
            sO[idx] = ((0.0f + -1.0f * a) + 0.0f * b) + 1.0f * c;
        }
    }
    __syncthreads();
    const double kc = octave_contrast(d_k[blockIdx.z], k_pow);
    const double inverse_k = 1.0 / (kc * kc);
    for (int idx = tid; idx < TH * TW; idx += NT) {  // V passes + pm_g2
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0 + lx, y = y0 + ly;
        if (x < w && y < h) {
            const int cx = clampi(x, 1, w - 2), cy = clampi(y, 1, h - 2);
            const int o = (cy - (y0 - 1)) * TW + (cx - x0);
            const float lx1 = ((0.0f + -1.0f * sM[o - TW]) + 0.0f * sM[o]) + 1.0f * sM[o + TW];
            const float ly1 = ((0.0f + kn * sO[o - TW]) + kwn * sO[o]) + kn * sO[o + TW];
            lflow[base + (size_t)y * w + x] = pm_g2_px(lx1, ly1, inverse_k);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Multiscale first derivatives (detector_response.rs:9-10): Lx = V_off(H_main(Ls)),
// Ly = V_main(H_off(Ls)) at scale S: taps at -S, 0, +S (the zero taps of the dense kernel add
// +-0 and are skipped).
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(NT)
k_deriv1(const float* __restrict__ ls, float* __restrict__ lx_out, float* __restrict__ ly_out, int w, int h,
         float kn, float kwn) {
    constexpr int RW = TW + 2 * S, RH = TH + 2 * S;
    __shared__ float sIn[RH * RW];
    __shared__ float sM[RH * TW];
    __shared__ float sO[RH * TW];
    const int tid = threadIdx.x;
    const int x0 = tile_origin(blockIdx.x, TW, w), y0 = tile_origin(blockIdx.y, TH, h);
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    for (int idx = tid; idx < RH * RW; idx += NT) {
        const int ly = idx / RW, lx = idx - ly * RW;
        const int gx = x0 - S + lx, gy = y0 - S + ly;
        sIn[idx] = (gx >= 0 && gx < w && gy >= 0 && gy < h) ? ls[base + (size_t)gy * w + gx] : 0.0f;
    }
    __syncthreads();
    for (int idx = tid; idx < RH * TW; idx += NT) {
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0 + lx, y = y0 - S + ly;
        if (x < w && y >= 0 && y < h) {
            const int cx = clampi(x, S, w - 1 - S), cy = clampi(y, S, h - 1 - S);
            const float* p = sIn + (cy - (y0 - S)) * RW + (cx - (x0 - S));
            const float a = p[-S], b = p[0], c = p[S];
            sM[idx] = ((0.0f + kn * a) + kwn * b) + kn * c;
            sO[idx] = ((0.0f + -1.0f * a) + 0.0f * b) + 1.0f * c;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < TH * TW; idx += NT) {
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0 + lx, y = y0 + ly;
        if (x < w && y < h) {
            const int cx = clampi(x, S, w - 1 - S), cy = clampi(y, S, h - 1 - S);
            const int o = (cy - (y0 - S)) * TW + (cx - x0);
            const size_t g = base + (size_t)y * w + x;
            lx_out[g] = ((0.0f + -1.0f * sM[o - S * TW]) + 0.0f * sM[o]) + 1.0f * sM[o + S * TW];
            ly_out[g] = ((0.0f + kn * sO[o - S * TW]) + kwn * sO[o]) + kn * sO[o + S * TW];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Second derivatives + Hessian determinant (detector_response.rs:11-13, :52):
// Lxx = V_off(H_main(Lx)), Lyy = V_main(H_off(Ly)), Lxy = V_main(H_off(Lx)),
// Ldet = ((Lxx*Lyy) - (Lxy*Lxy)) * sigma^4.  The second-derivative planes are written only if
// the caller keeps them.
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(NT)
k_deriv2(const float* __restrict__ lx_in, const float* __restrict__ ly_in, float* __restrict__ lxx_out,
         float* __restrict__ lyy_out, float* __restrict__ lxy_out, float* __restrict__ ldet_out, int w, int h,
         float kn, float kwn, float quat) {
    constexpr int RW = TW + 2 * S, RH = TH + 2 * S;
    __shared__ float sX[RH * RW];
    __shared__ float sY[RH * RW];
    __shared__ float sA[RH * TW];  // H_main(Lx)
    __shared__ float sB[RH * TW];  // H_off(Ly)
    __shared__ float sC[RH * TW];  // H_off(Lx)
    const int tid = threadIdx.x;
    const int x0 = tile_origin(blockIdx.x, TW, w), y0 = tile_origin(blockIdx.y, TH, h);
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    for (int idx = tid; idx < RH * RW; idx += NT) {
        const int ly = idx / RW, lx = idx - ly * RW;
        const int gx = x0 - S + lx, gy = y0 - S + ly;
        const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
        const size_t g = base + (size_t)gy * w + gx;
        sX[idx] = in ? lx_in[g] : 0.0f;
        sY[idx] = in ? ly_in[g] : 0.0f;
    }
    __syncthreads();
    for (int idx = tid; idx < RH * TW; idx += NT) {
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0 + lx, y = y0 - S + ly;
        if (x < w && y >= 0 && y < h) {
            const int cx = clampi(x, S, w - 1 - S), cy = clampi(y, S, h - 1 - S);
            const int o = (cy - (y0 - S)) * RW + (cx - (x0 - S));
            const float xa = sX[o - S], xb = sX[o], xc = sX[o + S];
            const float ya = sY[o - S], yb = sY[o], yc = sY[o + S];
            sA[idx] = ((0.0f + kn * xa) + kwn * xb) + kn * xc;
            sB[idx] = ((0.0f + -1.0f * ya) + 0.0f * yb) + 1.0f * yc;
            sC[idx] = ((0.0f + -1.0f * xa) + 0.0f * xb) + 1.0f * xc;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < TH * TW; idx += NT) {
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0 + lx, y = y0 + ly;
        if (x < w && y < h) {
            const int cx = clampi(x, S, w - 1 - S), cy = clampi(y, S, h - 1 - S);
            const int o = (cy - (y0 - S)) * TW + (cx - x0);
            const float lxx = ((0.0f + -1.0f * sA[o - S * TW]) + 0.0f * sA[o]) + 1.0f * sA[o + S * TW];
            const float lyy = ((0.0f + kn * sB[o - S * TW]) + kwn * sB[o]) + kn * sB[o + S * TW];
            const float lxy = ((0.0f + kn * sC[o - S * TW]) + kwn * sC[o]) + kn * sC[o + S * TW];
            const size_t g = base + (size_t)y * w + x;
            if (lxx_out) lxx_out[g] = lxx;
            if (lyy_out) lyy_out[g] = lyy;
            if (lxy_out) lxy_out[g] = lxy;
            ldet_out[g] = ((lxx * lyy) - (lxy * lxy)) * quat;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// k_deriv2 with the extrema test fused in: Ldet is evaluated on the tile plus a one-pixel ring
// (kept in LDS, aliased onto the dead Lx window), so the strict 4-neighbour maximum + threshold +
// descriptor-border test of scale_space_extrema.rs:32-42, :80-87 run without re-reading Ldet from
// HBM.  Candidates are appended unordered; the host sorts them into raster order.
// ---------------------------------------------------------------------------------------------
template <int S>
__global__ void __launch_bounds__(NT)
k_deriv2_nms(const float* __restrict__ lx_in, const float* __restrict__ ly_in, float* __restrict__ lxx_out,
             float* __restrict__ lyy_out, float* __restrict__ lxy_out, float* __restrict__ ldet_out, int w, int h,
             float kn, float kwn, float quat, unsigned level, float thr, float border_m,
             Candidate* __restrict__ cand, unsigned cap, unsigned* __restrict__ count) {
    constexpr int DW = TW + 2, DH = TH + 2;          // Ldet window, origin (x0-1, y0-1)
    constexpr int AW = DW, AH = DH + 2 * S;          // H windows,   origin (x0-1, y0-1-S)
    constexpr int RW = DW + 2 * S, RH = DH + 2 * S;  // input windows, origin (x0-1-S, y0-1-S)
    static_assert(DW * DH <= RH * RW, "Ldet window must fit in the Lx window it aliases");
    __shared__ float sX[RH * RW];
    __shared__ float sY[RH * RW];
    __shared__ float sA[AH * AW];  // H_main(Lx)
    __shared__ float sB[AH * AW];  // H_off(Ly)
    __shared__ float sC[AH * AW];  // H_off(Lx)
    float* sD = sX;                // Ldet window (valid after the second barrier)
    const int tid = threadIdx.x;
    const int x0 = tile_origin(blockIdx.x, TW, w), y0 = tile_origin(blockIdx.y, TH, h);
    const size_t base = (size_t)blockIdx.z * (size_t)w * (size_t)h;
    for (int idx = tid; idx < RH * RW; idx += NT) {
        const int ly = idx / RW, lx = idx - ly * RW;
        const int gx = x0 - 1 - S + lx, gy = y0 - 1 - S + ly;
        const bool in = gx >= 0 && gx < w && gy >= 0 && gy < h;
        const size_t g = base + (size_t)gy * w + gx;
        sX[idx] = in ? lx_in[g] : 0.0f;
        sY[idx] = in ? ly_in[g] : 0.0f;
    }
    __syncthreads();
    for (int idx = tid; idx < AH * AW; idx += NT) {
        const int ly = idx / AW, lx = idx - ly * AW;
        const int x = x0 - 1 + lx, y = y0 - 1 - S + ly;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            const int cx = clampi(x, S, w - 1 - S), cy = clampi(y, S, h - 1 - S);
            const int o = (cy - (y0 - 1 - S)) * RW + (cx - (x0 - 1 - S));
            const float xa = sX[o - S], xb = sX[o], xc = sX[o + S];
            const float ya = sY[o - S], yb = sY[o], yc = sY[o + S];
            sA[idx] = ((0.0f + kn * xa) + kwn * xb) + kn * xc;
            sB[idx] = ((0.0f + -1.0f * ya) + 0.0f * yb) + 1.0f * yc;
            sC[idx] = ((0.0f + -1.0f * xa) + 0.0f * xb) + 1.0f * xc;
        }
    }
    __syncthreads();  // sX / sY are dead from here on
    for (int idx = tid; idx < DH * DW; idx += NT) {
        const int ly = idx / DW, lx = idx - ly * DW;
        const int x = x0 - 1 + lx, y = y0 - 1 + ly;
        if (x >= 0 && x < w && y >= 0 && y < h) {
            const int cx = clampi(x, S, w - 1 - S), cy = clampi(y, S, h - 1 - S);
            const int o = (cy - (y0 - 1 - S)) * AW + (cx - (x0 - 1));
            const float lxx = ((0.0f + -1.0f * sA[o - S * AW]) + 0.0f * sA[o]) + 1.0f * sA[o + S * AW];
            const float lyy = ((0.0f + kn * sB[o - S * AW]) + kwn * sB[o]) + kn * sB[o + S * AW];
            const float lxy = ((0.0f + kn * sC[o - S * AW]) + kwn * sC[o]) + kn * sC[o + S * AW];
            const float det = ((lxx * lyy) - (lxy * lxy)) * quat;
            sD[idx] = det;
            if (lx >= 1 && lx <= TW && ly >= 1 && ly <= TH) {
                const size_t g = base + (size_t)y * w + x;
                if (lxx_out) lxx_out[g] = lxx;
                if (lyy_out) lyy_out[g] = lyy;
                if (lxy_out) lxy_out[g] = lxy;
                ldet_out[g] = det;
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < TH * TW; idx += NT) {
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = x0 + lx, y = y0 + ly;
        // flat range (w+1) .. len-w-2 of the reference loop; x = w-1 never passes the border test
        if (x < 1 || x > w - 2 || y < 1 || y > h - 2) continue;
        if ((long)y * w + x >= (long)w * h - w - 1) continue;
        const int o = (ly + 1) * DW + (lx + 1);
        const float v = sD[o];
        if (!(v > thr)) continue;
        const float xp = sD[o + 1], xm = sD[o - 1], yp = sD[o + DW], ym = sD[o - DW];
        if (!(v > xp && v > xm && v > ym && v > yp)) continue;
        const float fx = (float)x, fy = (float)y;
        const bool is_out = (roundf(fx - border_m) - 1.0f) < 0.0f || (roundf(fx + border_m) + 1.0f) >= (float)w ||
                            (roundf(fy - border_m) - 1.0f) < 0.0f || (roundf(fy + border_m) + 1.0f) >= (float)h;
        if (is_out) continue;
        // tiles shifted inward overlap their neighbour: only the owner of a pixel reports it
        if (x < (int)blockIdx.x * TW || y < (int)blockIdx.y * TH) continue;
        const unsigned slot = atomicAdd(count, 1u);
        if (slot < cap) {
            Candidate c;
            c.level = level;
            c.idx = (unsigned)(y * w + x);
            c.v = v; c.xp = xp; c.xm = xm; c.yp = yp; c.ym = ym;
            c.img = blockIdx.z;
            cand[slot] = c;
        }
    }
}

inline dim3 tiles(uint32_t w, uint32_t h, uint32_t n) { return dim3((w + TW - 1) / TW, (h + TH - 1) / TH, n); }

}  // namespace

namespace launch {

bool blur_fused_supported(uint32_t ntaps) { return ntaps == 3 || ntaps == 5; }

template <typename T>
static void blur_fused_t(hipStream_t s, const T* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k,
                         uint32_t ntaps) {
    DenseTaps t;
    for (uint32_t i = 0; i < (uint32_t)kMaxTaps; ++i) t.k[i] = i < ntaps ? k[i] : 0.0f;
    if (ntaps == 3)
        hipLaunchKernelGGL((k_blur<1, T>), tiles(w, h, n), dim3(NT), 0, s, in, out, (int)w, (int)h, t);
    else
        hipLaunchKernelGGL((k_blur<2, T>), tiles(w, h, n), dim3(NT), 0, s, in, out, (int)w, (int)h, t);
}
void blur_fused_f32(hipStream_t s, const float* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k,
                    uint32_t ntaps) {
    blur_fused_t<float>(s, in, out, w, h, n, k, ntaps);
}
void blur_fused_u8(hipStream_t s, const uint8_t* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k,
                   uint32_t ntaps) {
    blur_fused_t<uint8_t>(s, in, out, w, h, n, k, ntaps);
}

void prep_fused(hipStream_t s, const float* prev, bool half, float* lt_out, float* lsmooth, float* lflow, uint32_t w,
                uint32_t h, uint32_t pw, uint32_t ph, uint32_t n, const float* g3, const double* d_k,
                uint32_t k_pow) {
    const Taps m = taps_scharr_main(1);
    if (half)
        hipLaunchKernelGGL((k_prep<true>), tiles(w, h, n), dim3(NT), 0, s, prev, lt_out, lsmooth, lflow, (int)w,
                           (int)h, (int)pw, (int)ph, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow);
    else
        hipLaunchKernelGGL((k_prep<false>), tiles(w, h, n), dim3(NT), 0, s, prev, lt_out, lsmooth, lflow, (int)w,
                           (int)h, (int)pw, (int)ph, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow);
}

bool detector_fused_supported(uint32_t sigma) { return sigma >= 1 && sigma <= 6; }
bool detector_nms_fused_supported(uint32_t sigma) { return sigma >= 1 && sigma <= 4; }  // LDS <= 64 KB

void detector_nms_fused(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx,
                        float* lyy, float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n, uint32_t level,
                        float thr, float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count) {
    const Taps m = taps_scharr_main(sigma);
    const float kn = m.wgt[0], kwn = m.wgt[1];
    const float quat = (float)(sigma * sigma * sigma * sigma);
    const dim3 g = tiles(w, h, n);
#define AKZ_DETN(S)                                                                                                \
    case S:                                                                                                        \
        hipLaunchKernelGGL((k_deriv1<S>), g, dim3(NT), 0, s, lsmooth, lx, ly, (int)w, (int)h, kn, kwn);            \
        hipLaunchKernelGGL((k_deriv2_nms<S>), g, dim3(NT), 0, s, (const float*)lx, (const float*)ly, lxx, lyy, lxy, \
                           ldet_out, (int)w, (int)h, kn, kwn, quat, level, thr, border_m, d_cand, cap, d_count);    \
        break;
    switch (sigma) {
        AKZ_DETN(1) AKZ_DETN(2) AKZ_DETN(3) AKZ_DETN(4)
        default: break;
    }
#undef AKZ_DETN
}

void detector_fused(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx, float* lyy,
                    float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n) {
    const Taps m = taps_scharr_main(sigma);
    const float kn = m.wgt[0], kwn = m.wgt[1];
    const float quat = (float)(sigma * sigma * sigma * sigma);
    const dim3 g = tiles(w, h, n);
#define AKZ_DET(S)                                                                                              \
    case S:                                                                                                     \
        hipLaunchKernelGGL((k_deriv1<S>), g, dim3(NT), 0, s, lsmooth, lx, ly, (int)w, (int)h, kn, kwn);         \
        hipLaunchKernelGGL((k_deriv2<S>), g, dim3(NT), 0, s, (const float*)lx, (const float*)ly, lxx, lyy, lxy, \
                           ldet_out, (int)w, (int)h, kn, kwn, quat);                                            \
        break;
    switch (sigma) {
        AKZ_DET(1) AKZ_DET(2) AKZ_DET(3) AKZ_DET(4) AKZ_DET(5) AKZ_DET(6)
        default: break;
    }
#undef AKZ_DET
}

}  // namespace launch
}  // namespace akz
