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
// Workgroups are PERSISTENT: a workgroup walks over tiles (grid-stride over x, y and the image of
// the batch) and issues the global loads of the NEXT tile's input window into registers before it
// runs the passes of the current tile, so the HBM latency of one tile hides under the LDS work of
// the previous one (these kernels do little work per byte and were latency-bound without it).
//
// Arithmetic is the reference's: f32 mul then add, taps left to right starting from 0.0f, no FMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "akz_internal.hpp"
#include "akz_pm_g2.hpp"
#include <type_traits>

#include "akz_prep_passes.hpp"

namespace akz {
namespace {

#ifndef AKZ_PERSIST_BLOCKS
#define AKZ_PERSIST_BLOCKS 1024  // 256 CUs x 4 workgroups of 512 threads
#endif
constexpr int TW = 64, TH = 32, NT = 512;
// k_deriv2 keeps the common tile height: 64 x 30 / 64 x 28 tiles (fewer loads per thread at sigma 4) measured the same
constexpr int deriv2_tile_h(int) { return TH; }
// k_detector_tiled: its two LDS buffers grow with sigma_size -- 51.4 / 54.7 / 58.1 KB at S = 2 / 3 / 4 with 64 x 32 tiles, i.e.
// TWO workgroups per CU from S = 3 on (160 KB of LDS).  Tiles of 30 / 28 rows keep them under 53.3 KB: three per CU.
constexpr int det_tile_h(int S) { return S <= 2 ? TH : (S == 3 ? 30 : 28); }

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// origin of tile b along a dimension of `dim` pixels
__device__ __forceinline__ int tile_origin(int b, int tile, int dim) {
    const int o = b * tile;
    return (dim > tile && o + tile > dim) ? dim - tile : o;
}

struct TileGrid {
    int tx, ty, n;  // tiles per row / column, images
};
struct Tile {
    int bx, by, bz, x0, y0;
};
template <int TILE_H = TH>
__device__ __forceinline__ Tile decode_tile(int t, TileGrid g, int w, int h) {
    const int per = g.tx * g.ty;
    Tile r;
    r.bz = t / per;
    const int rem = t - r.bz * per;
    r.by = rem / g.tx;
    r.bx = rem - r.by * g.tx;
    r.x0 = tile_origin(r.bx, TW, w);
    r.y0 = tile_origin(r.by, TILE_H, h);
    return r;
}

__device__ __forceinline__ float unit_px(const float* p, size_t i) { return p[i]; }
// create_unit_float_image: f32::from(v) * 1f32 / 255f32 (types/image.rs:136)
__device__ __forceinline__ float unit_px(const uint8_t* p, size_t i) { return ((float)p[i] * 1.0f) / 255.0f; }

__device__ __forceinline__ double octave_contrast(double k, unsigned pow) {
    for (unsigned i = 0; i < pow; ++i) k = k * 0.75;  // lib.rs:84, one octave at a time in f64
    return k;
}

struct DenseTaps {
    float k[kMaxTaps];
};

// ---------------------------------------------------------------------------------------------
// gaussian_blur = V(H(in)) with a dense (2*HW+1)-tap kernel (types/image.rs:374-380)
// ---------------------------------------------------------------------------------------------
template <int HW, typename T>
__global__ void __launch_bounds__(NT)
k_blur(const T* __restrict__ in, float* __restrict__ out, int w, int h, TileGrid tg, DenseTaps t) {
    constexpr int RW = TW + 2 * HW, RH = TH + 2 * HW;
    constexpr int NLOAD = (RH * RW + NT - 1) / NT;
    __shared__ float sIn[RH * RW];
    __shared__ float sH[RH * TW];
    const int tid = threadIdx.x;
    const int ntiles = tg.tx * tg.ty * tg.n;
    float regs[NLOAD];
    auto issue = [&](int tile) {
        const Tile tl = decode_tile(tile, tg, w, h);
        const size_t base = (size_t)tl.bz * (size_t)w * (size_t)h;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * NT;
            const int ly = idx / RW, lx = idx - ly * RW;
            const int gx = tl.x0 - HW + lx, gy = tl.y0 - HW + ly;
            regs[k] = (idx < RH * RW && gx >= 0 && gx < w && gy >= 0 && gy < h)
                          ? unit_px(in, base + (size_t)gy * w + gx) : 0.0f;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * NT;
            if (idx < RH * RW) sIn[idx] = regs[k];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);  // in flight during the passes below
        const Tile tl = decode_tile(tile, tg, w, h);
        const int x0 = tl.x0, y0 = tl.y0;
        const size_t base = (size_t)tl.bz * (size_t)w * (size_t)h;
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
        __syncthreads();  // the windows are overwritten by the next iteration
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
       float* __restrict__ lflow, int w, int h, int pw, int ph, TileGrid tg, float g0, float g1, float g2, float kn,
       float kwn, const double* __restrict__ d_k, unsigned k_pow) {
    constexpr int IW = TW + 4, IH = TH + 4;  // input window, origin (x0-2, y0-2)
    constexpr int AW = TW + 2, AH = TH + 4;  // H_g,       origin (x0-1, y0-2)
    constexpr int BW = TW + 2, BH = TH + 2;  // Lsmooth,   origin (x0-1, y0-1)
    constexpr int CH = TH + 2;               // H_scharr,  origin (x0,   y0-1), width TW
    constexpr int NLOAD = (IH * IW + NT - 1) / NT;
    // The Scharr H-pass windows take the place of the windows that are dead by then (the input window after the first pass,
    // H_g after the second; a barrier lies between every pass): 28.3 KB instead of 45.7, i.e. FOUR 512-thread workgroups per
    // CU instead of three -- the 1 020 tiles of a lone 1080p level run as one round instead of one and a third.
    static_assert(CH * TW <= IH * IW && CH * TW <= AH * AW, "the Scharr windows fit the windows they alias");
    __shared__ float sI[IH * IW];
    __shared__ float sA[AH * AW];
    __shared__ float sB[BH * BW];
    const int tid = threadIdx.x;
    const int ntiles = tg.tx * tg.ty * tg.n;
    float regs[NLOAD];
    auto issue = [&](int tile) {
        const Tile tl = decode_tile(tile, tg, w, h);
        const float* src = prev + (size_t)tl.bz * (size_t)pw * (size_t)ph;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * NT;
            const int ly = idx / IW, lx = idx - ly * IW;
            const int gx = tl.x0 - 2 + lx, gy = tl.y0 - 2 + ly;
            float v = 0.0f;
            if (idx < IH * IW && gx >= 0 && gx < w && gy >= 0 && gy < h) {
                if (HALF) {
                    const float* q = src + (size_t)(2 * gy) * pw + 2 * gx;
                    v = v + q[0];
                    v = v + q[pw];
                    v = v + q[1];
                    v = v + q[pw + 1];
                    v = v / 4.0f;
                } else {
                    v = src[(size_t)gy * w + gx];
                }
            }
            regs[k] = v;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const Tile tl = decode_tile(tile, tg, w, h);
        const int x0 = tl.x0, y0 = tl.y0;
        const size_t base = (size_t)tl.bz * (size_t)w * (size_t)h;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * NT;
            if (idx < IH * IW) {
                sI[idx] = regs[k];
                if (HALF) {  // the 2x2 mean of the centre is this level's starting Lt
                    const int ly = idx / IW, lx = idx - ly * IW;
                    const int gx = x0 - 2 + lx, gy = y0 - 2 + ly;
                    if (lx >= 2 && lx < TW + 2 && ly >= 2 && ly < TH + 2 && gx < w && gy < h)
                        lt_out[base + (size_t)gy * w + gx] = regs[k];
                }
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        // the four passes on the window (akz_prep_passes.hpp: shared with the diffusion kernel's epilogue)
        const double kc = octave_contrast(d_k[tl.bz], k_pow);
        prep_passes<TW, TH, NT>(sI, sA, sB, x0, y0, w, h, base, lsmooth, lflow, PrepTaps{g0, g1, g2, kn, kwn}, 1.0 / (kc * kc));
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Level 0 of a small job in ONE launch (lib.rs:56-60, contrast_factor.rs:18-40): the frame -> Lt0 = gaussian_blur(frame,
// base_scale_offset) (5 taps, stored) -> the contrast factor's own gaussian_blur(Lt0, 1.0) (3 taps, stored for the histogram
// pass) -> its scale-1 Scharr pair (stored: the histogram pass bins it, and level 1 -- whose Lsmooth is this very blur of Lt0,
// lib.rs:92-95 -- forms its Lflow from it) -> the LARGEST squared gradient magnitude of the image (f64, one atomicMax per
// workgroup and image).  Four launches before -- blur, the clearing of the contrast scratch, blur, k_contrast_max: 44 us of a lone
// 1080p frame's 760 -- each a few microseconds of work behind its launch's floor.
// The maximum is taken over dx*dx + dy*dy and its square root once (by the histogram pass): sqrt is monotone and correctly
// rounded, so sqrt(max s) IS max sqrt(s), the reference's hmax, and no pixel pays for an f64 square root here.  Border pixels
// enter with their clamped (interior) coordinates, i.e. as duplicates of interior pixels (the reference walks the interior
// only, :27-38).
// Windows: frame +-4, H pass of the 5-tap blur rows +-4 cols +-2, Lt0 +-2 -- which is the input window of the preparation's
// passes (akz_prep_passes.hpp), whose Gaussian and Scharr stages are the contrast factor's (the same kernel sizes).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(NT)
k_head(const T* __restrict__ in, float* __restrict__ lt0, float* __restrict__ blurred, float* __restrict__ gx_out, float* __restrict__ gy_out,
       int w, int h, TileGrid tg, DenseTaps t5, PrepTaps t3, unsigned long long* __restrict__ d_smax_bits, unsigned* __restrict__ zero_word) {
    constexpr int IW = TW + 8, IH = TH + 8;  // frame,  origin (x0-4, y0-4)
    constexpr int HW = TW + 4, HH = TH + 8;  // H pass, origin (x0-2, y0-4)
    constexpr int LW = TW + 4, LH = TH + 4;  // Lt0,    origin (x0-2, y0-2)
    constexpr int NLOAD = (IH * IW + NT - 1) / NT;
    static_assert((TH + 4) * (TW + 2) <= IH * IW && (TH + 2) * (TW + 2) <= HH * HW, "the preparation's scratch windows fit the dead ones");
    __shared__ float sIn[IH * IW];  // (then the passes' H_g window)
    __shared__ float sH[HH * HW];   // (then the passes' Lsmooth window)
    __shared__ float sL[LH * LW];
    __shared__ unsigned long long s_part[NT / 64];
    const int tid = threadIdx.x;
    const int ntiles = tg.tx * tg.ty * tg.n;
    if (zero_word && blockIdx.x == 0 && tid == 0) *zero_word = 0u;  // (the job's candidate counter: its first user is a later launch)
    float regs[NLOAD];
    auto issue = [&](int tile) {
        const Tile tl = decode_tile(tile, tg, w, h);
        const size_t base = (size_t)tl.bz * (size_t)w * (size_t)h;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * NT;
            const int ly = idx / IW, lx = idx - ly * IW;
            const int gx = tl.x0 - 4 + lx, gy = tl.y0 - 4 + ly;
            regs[k] = (idx < IH * IW && gx >= 0 && gx < w && gy >= 0 && gy < h) ? unit_px(in, base + (size_t)gy * w + gx) : 0.0f;
        }
    };
    double m = 0.0;  // this thread's largest dx*dx + dy*dy of image m_img
    int m_img = -1;
    auto flush = [&](int img) {  // (all threads; non-negative doubles order like their bit patterns)
        unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(bits, o, 64);
            bits = other > bits ? other : bits;
        }
        if ((tid & 63) == 0) s_part[tid >> 6] = bits;
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < NT / 64; ++i) bits = s_part[i] > bits ? s_part[i] : bits;
            if (bits != 0ull) atomicMax(d_smax_bits + img, bits);
        }
        __syncthreads();
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        const Tile tl = decode_tile(tile, tg, w, h);
        const int x0 = tl.x0, y0 = tl.y0;
        const size_t base = (size_t)tl.bz * (size_t)w * (size_t)h;
        if (tl.bz != m_img) {  // (workgroup-uniform)
            if (m_img >= 0) flush(m_img);
            m_img = tl.bz;
            m = 0.0;
        }
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * NT;
            if (idx < IH * IW) sIn[idx] = regs[k];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        const bool interior = x0 >= 6 && x0 + TW + 6 <= w && y0 >= 6 && y0 + TH + 6 <= h;  // (no test, no clamp: as in k_detector_tiled)
        auto blur5 = [&](auto in_tag) {
            constexpr bool IN = decltype(in_tag)::value;
            // (a thread owns a window column and walks down it, as in k_detector_tiled)
            {  // H pass of the 5-tap blur (types/image.rs:374-380), filled
                constexpr int RS = NT / HW;
                const int r0 = tid / HW, lx = tid - r0 * HW;
                const int x = x0 - 2 + lx;
                if (r0 < RS && (IN || (x >= 0 && x < w))) {
                    const int cx = IN ? x : clampi(x, 2, w - 3);
                    for (int ly = r0; ly < HH; ly += RS) {
                        const int y = y0 - 4 + ly;
                        if (!IN && (y < 0 || y >= h)) continue;
                        const int cy = IN ? y : clampi(y, 2, h - 3);
                        const float* p = sIn + (cy - (y0 - 4)) * IW + (cx - 2 - (x0 - 4));
                        float acc = 0.0f;
#pragma unroll
                        for (int i = 0; i < 5; ++i) acc = acc + t5.k[i] * p[i];
                        sH[ly * HW + lx] = acc;
                    }
                }
            }
            __syncthreads();
            {  // V pass: Lt0 on the tile + 2
                constexpr int RS = NT / LW;
                const int r0 = tid / LW, lx = tid - r0 * LW;
                const int x = x0 - 2 + lx;
                if (r0 < RS && (IN || (x >= 0 && x < w))) {
                    const int cx = IN ? x : clampi(x, 2, w - 3);
                    const bool own_x = lx >= 2 && lx < TW + 2;
                    for (int ly = r0; ly < LH; ly += RS) {
                        const int y = y0 - 2 + ly;
                        if (!IN && (y < 0 || y >= h)) continue;
                        const int cy = IN ? y : clampi(y, 2, h - 3);
                        const float* p = sH + (cy - 2 - (y0 - 4)) * HW + (cx - (x0 - 2));
                        float acc = 0.0f;
#pragma unroll
                        for (int i = 0; i < 5; ++i) acc = acc + t5.k[i] * p[i * HW];
                        sL[ly * LW + lx] = acc;
                        if (own_x && ly >= 2 && ly < TH + 2) lt0[base + (size_t)y * w + x] = acc;
                    }
                }
            }
            __syncthreads();
        };
        if (interior) blur5(std::true_type{});
        else blur5(std::false_type{});
        prep_passes_fin<TW, TH, NT>(sL, sIn, sH, x0, y0, w, h, base, blurred, t3, [&](int x, int y, float lx1, float ly1) {
            const size_t gi = base + (size_t)y * w + x;  // the (filled) Scharr pair: the histogram pass and level 1's Lflow read it
            gx_out[gi] = lx1;
            gy_out[gi] = ly1;
            const double dx = (double)lx1, dy = (double)ly1;
            const double s2 = dx * dx + dy * dy;
            if (s2 > m) m = s2;
        });
        __syncthreads();
    }
    if (m_img >= 0) flush(m_img);
}

// ---------------------------------------------------------------------------------------------
// Quad helpers: the detector kernels process four consecutive pixels of a row per thread.  Window
// origins and pitches are multiples of 4 floats, so the centre taps are one aligned float4 and
// vertical taps are aligned float4s of other rows; horizontal taps at -S / +S are assembled from the
// neighbouring elements.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
// a = p[-S .. -S+3], b = p[0..3], c = p[S .. S+3]; p is 16-byte aligned
template <int S>
__device__ __forceinline__ void h_taps(const float* p, float (&a)[4], float (&b)[4], float (&c)[4]) {
    ld4(p, b);
    if (S == 4) {
        ld4(p - 4, a);
        ld4(p + 4, c);
    } else if (S == 2) {
        const float2 l = *reinterpret_cast<const float2*>(p - 2), r = *reinterpret_cast<const float2*>(p + 4);
        a[0] = l.x; a[1] = l.y; a[2] = b[0]; a[3] = b[1];
        c[0] = b[2]; c[1] = b[3]; c[2] = r.x; c[3] = r.y;
    } else if (S == 1) {
        a[0] = p[-1]; a[1] = b[0]; a[2] = b[1]; a[3] = b[2];
        c[0] = b[1]; c[1] = b[2]; c[2] = b[3]; c[3] = p[4];
    } else if (S == 3) {
        a[0] = p[-3]; a[1] = p[-2]; a[2] = p[-1]; a[3] = b[0];
        c[0] = b[3]; c[1] = p[4]; c[2] = p[5]; c[3] = p[6];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = p[i - S];
            c[i] = p[i + S];
        }
    }
}
// the two separable Scharr taps sets (derivatives.rs:74-101), evaluated left to right from 0.0f
__device__ __forceinline__ float tap_main(float a, float b, float c, float kn, float kwn) {
    return ((0.0f + kn * a) + kwn * b) + kn * c;
}
// off-axis taps [-1, 0.., 0, ..0, 1]: ((0.0f + -1.0f*a) + 0.0f*b) + 1.0f*c == (0.0f - a) + c bit for bit for every
// finite b (the zero taps add +-0 to an accumulator that is never -0); four VALU operations fewer per value
__device__ __forceinline__ float tap_off(float a, float /*b*/, float c) {
    return (0.0f - a) + c;
}
// global load of 4 consecutive pixels of row gy starting at gx (zeros outside the image)
__device__ __forceinline__ float4 load_quad(const float* __restrict__ plane, int gx, int gy, int w, int h, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy < 0 || gy >= h || gx + 3 < 0 || gx >= w) return v;
    const float* row = plane + (size_t)gy * w;
    if (vec_ok && gx >= 0 && gx + 3 < w) return *reinterpret_cast<const float4*>(row + gx);
    float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (gx + e >= 0 && gx + e < w) t[e] = row[gx + e];
    return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ void store_quad(float* __restrict__ plane, int gx, int gy, int w, bool vec_ok,
                                           const float (&v)[4]) {
    float* row = plane + (size_t)gy * w;
    if (vec_ok && gx + 3 < w) {
        plane_store4(row + gx, v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (gx + e < w) plane_store(row + gx + e, v[e]);
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
         TileGrid tg, float kn, float kwn) {
    constexpr int HX = (S + 3) / 4 * 4;          // x halo of the input window, multiple of 4
    constexpr int RW = TW + 2 * HX, RH = TH + 2 * S;
    constexpr int RQ = RW / 4, TQ = TW / 4;      // quads per row
    constexpr int NLOAD = (RH * RQ + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float sIn[RH * RW];
    __shared__ __attribute__((aligned(16))) float sM[RH * TW];
    __shared__ __attribute__((aligned(16))) float sO[RH * TW];
    const int tid = threadIdx.x;
    const int ntiles = tg.tx * tg.ty * tg.n;
    const bool vec_ok = (w & 3) == 0;
    float4 regs[NLOAD];
    auto issue = [&](int tile) {
        const Tile tl = decode_tile(tile, tg, w, h);
        const float* src = ls + (size_t)tl.bz * (size_t)w * (size_t)h;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int q = tid + k * NT;
            const int ly = q / RQ, qc = q - ly * RQ;
            regs[k] = q < RH * RQ ? load_quad(src, tl.x0 - HX + 4 * qc, tl.y0 - S + ly, w, h, vec_ok)
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int q = tid + k * NT;
            if (q < RH * RQ) *reinterpret_cast<float4*>(sIn + 4 * q) = regs[k];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        const Tile tl = decode_tile(tile, tg, w, h);
        const int x0 = tl.x0, y0 = tl.y0;
        float* lxp = lx_out + (size_t)tl.bz * (size_t)w * (size_t)h;
        float* lyp = ly_out + (size_t)tl.bz * (size_t)w * (size_t)h;
        for (int q = tid; q < RH * TQ; q += NT) {  // H passes on window rows y0-S .. y0+TH+S
            const int ly = q / TQ, qc = q - ly * TQ;
            const int x = x0 + 4 * qc, y = y0 - S + ly;
            if (y < 0 || y >= h || x >= w) continue;
            const int cy = clampi(y, S, h - 1 - S);
            const float* rowp = sIn + (cy - (y0 - S)) * RW + HX;  // element of column x0
            float hm[4], ho[4];
            if (x >= S && x + 3 <= w - 1 - S) {
                float a[4], b[4], c[4];
                h_taps<S>(rowp + 4 * qc, a, b, c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    hm[i] = tap_main(a[i], b[i], c[i], kn, kwn);
                    ho[i] = tap_off(a[i], b[i], c[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int cx = clampi(x + i, S, w - 1 - S);
                    const float* p = rowp + (cx - x0);
                    hm[i] = tap_main(p[-S], p[0], p[S], kn, kwn);
                    ho[i] = tap_off(p[-S], p[0], p[S]);
                }
            }
            st4(sM + 4 * q, hm);
            st4(sO + 4 * q, ho);
        }
        __syncthreads();
        for (int q = tid; q < TH * TQ; q += NT) {  // V passes on the centre
            const int ly = q / TQ, qc = q - ly * TQ;
            const int x = x0 + 4 * qc, y = y0 + ly;
            if (y >= h || x >= w) continue;
            const int cy = clampi(y, S, h - 1 - S);
            const int rowo = (cy - (y0 - S)) * TW;
            float vx[4], vy[4];
            if (x >= S && x + 3 <= w - 1 - S) {
                float m0[4], m1[4], m2[4], o0[4], o1[4], o2[4];
                ld4(sM + rowo - S * TW + 4 * qc, m0); ld4(sM + rowo + 4 * qc, m1); ld4(sM + rowo + S * TW + 4 * qc, m2);
                ld4(sO + rowo - S * TW + 4 * qc, o0); ld4(sO + rowo + 4 * qc, o1); ld4(sO + rowo + S * TW + 4 * qc, o2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    vx[i] = tap_off(m0[i], m1[i], m2[i]);
                    vy[i] = tap_main(o0[i], o1[i], o2[i], kn, kwn);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = rowo + (clampi(x + i, S, w - 1 - S) - x0);
                    vx[i] = tap_off(sM[o - S * TW], sM[o], sM[o + S * TW]);
                    vy[i] = tap_main(sO[o - S * TW], sO[o], sO[o + S * TW], kn, kwn);
                }
            }
            store_quad(lxp, x, y, w, vec_ok, vx);
            store_quad(lyp, x, y, w, vec_ok, vy);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Second derivatives + Hessian determinant (detector_response.rs:11-13, :52):
// Lxx = V_off(H_main(Lx)), Lyy = V_main(H_off(Ly)), Lxy = V_main(H_off(Lx)),
// Ldet = ((Lxx*Lyy) - (Lxy*Lxy)) * sigma^4, optionally with the extrema test fused in (NMS):
// Ldet is then evaluated on the tile plus a one-pixel ring (kept in LDS, aliased onto the dead Lx
// window), so the strict 4-neighbour maximum + threshold + descriptor-border test of
// scale_space_extrema.rs:32-42, :80-87 run without re-reading Ldet from HBM.  Candidates are appended
// unordered to one list; the host sorts them into raster order.  The second-derivative planes are
// written only if the caller keeps them.  (One pixel per thread and stage: a float4-per-thread form
// measured 3-4 % slower here — too few work items per phase for 512 threads.)
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Extrema of one tile from its Ldet window in LDS (window origin one pixel up-left of the tile, pitch DW):
// threshold + strict 4-neighbour maximum + descriptor-border test (scale_space_extrema.rs:32-42, :80-87).
// Slots of the global candidate list are reserved ONCE PER TILE: the threads count their extrema in an LDS
// counter, one thread adds the tile's total to the global counter (a frame with fine texture has several extrema
// per tile, and a million same-address device atomics per batch cost 0.6 ms), then every thread writes its
// candidates into the reserved range.  Candidates are appended unordered; the host sorts them into raster order.
// sCnt[0] must be zero on entry (thread 0 clears it behind an earlier barrier of the tile loop); the caller
// places a barrier after the call before the Ldet window is overwritten.
// ---------------------------------------------------------------------------------------------
template <int DW, int TILE_H = TH>
__device__ __forceinline__ void tile_extrema(const float* sD, unsigned* sCnt, int tid, const Tile& tl, int w, int h,
                                             float thr, float bm, unsigned level, Candidate* __restrict__ cand,
                                             unsigned cap, unsigned* __restrict__ count) {
    constexpr int IT = (TW * TILE_H + NT - 1) / NT;
    static_assert(IT <= 32, "one mask bit per pixel of a thread");
    unsigned mask = 0;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int idx = tid + i * NT;
        if (IT * NT != TW * TILE_H && idx >= TW * TILE_H) continue;
        const int ly = idx / TW, lx = idx - ly * TW;
        const int x = tl.x0 + lx, y = tl.y0 + ly;
        // flat range (w+1) .. len-w-2 of the reference loop; x = w-1 never passes the border test
        if (x < 1 || x > w - 2 || y < 1 || y > h - 2) continue;
        if ((long)y * w + x >= (long)w * h - w - 1) continue;
        const int o = (ly + 1) * DW + (lx + 1);
        const float v = sD[o];
        if (!(v > thr)) continue;
        if (!(v > sD[o + 1] && v > sD[o - 1] && v > sD[o - DW] && v > sD[o + DW])) continue;
        const float fx = (float)x, fy = (float)y;
        const bool is_out = (roundf(fx - bm) - 1.0f) < 0.0f || (roundf(fx + bm) + 1.0f) >= (float)w ||
                            (roundf(fy - bm) - 1.0f) < 0.0f || (roundf(fy + bm) + 1.0f) >= (float)h;
        if (is_out) continue;
        // tiles shifted inward overlap their neighbour: only the owner of a pixel reports it
        if (x < tl.bx * TW || y < tl.by * TILE_H) continue;
        mask |= 1u << i;
    }
    const unsigned cnt = (unsigned)__popc(mask);
    unsigned mine = 0;
    if (cnt) mine = atomicAdd(&sCnt[0], cnt);
    __syncthreads();
    if (sCnt[0] == 0) return;  // uniform
    if (tid == 0) sCnt[1] = atomicAdd(count, sCnt[0]);
    __syncthreads();
    unsigned slot = sCnt[1] + mine;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        if (!((mask >> i) & 1u)) continue;
        const int idx = tid + i * NT;
        const int ly = idx / TW, lx = idx - ly * TW;
        const int o = (ly + 1) * DW + (lx + 1);
        if (slot < cap) {
            Candidate c;
            c.level = level;
            c.idx = (unsigned)((tl.y0 + ly) * w + tl.x0 + lx);
            c.v = sD[o]; c.xp = sD[o + 1]; c.xm = sD[o - 1]; c.yp = sD[o + DW]; c.ym = sD[o - DW];
            c.img = (unsigned)tl.bz;
            cand[slot] = c;
        }
        ++slot;
    }
}

struct NmsArgs {
    unsigned level;
    float thr, border_m;
    Candidate* cand;
    unsigned cap;
    unsigned* count;
};

template <int S, bool NMS>
__global__ void __launch_bounds__(NT)
k_deriv2(const float* __restrict__ lx_in, const float* __restrict__ ly_in, float* __restrict__ lxx_out,
         float* __restrict__ lyy_out, float* __restrict__ lxy_out, float* __restrict__ ldet_out, int w, int h,
         TileGrid tg, float kn, float kwn, float quat, NmsArgs nms) {
    constexpr int T2 = deriv2_tile_h(S);                   // tile height of this kernel (see the defines)
    constexpr int RING = NMS ? 1 : 0;
    constexpr int DW = TW + 2 * RING, DH = T2 + 2 * RING;  // Ldet window, origin (x0-RING, y0-RING)
    constexpr int AW = DW, AH = DH + 2 * S;                // H windows,   origin (x0-RING, y0-RING-S)
    constexpr int RW = DW + 2 * S, RH = DH + 2 * S;        // input windows, origin (x0-RING-S, y0-RING-S)
    constexpr int NLOAD = (RH * RW + NT - 1) / NT;
    constexpr int NROWT = NT / TW;                         // thread rows of the 2-D mapping (8)
    static_assert(TW == 64 && NT % TW == 0, "one thread per tile column");
    static_assert(DW * DH <= RH * RW, "Ldet window must fit in the Lx window it aliases");
    __shared__ float sX[RH * RW];
    __shared__ float sY[RH * RW];
    __shared__ float sA[AH * AW];  // H_main(Lx)
    __shared__ float sB[AH * AW];  // H_off(Ly)
    __shared__ float sC[AH * AW];  // H_off(Lx)
    __shared__ unsigned sCnt[2];   // extrema of the tile, base slot in the candidate list
    float* sD = sX;                // Ldet window (valid after the second barrier)
    const int tid = threadIdx.x;
    const int tx = tid & (TW - 1), ty = tid / TW;
    const int ntiles = tg.tx * tg.ty * tg.n;
    // window coordinates of this thread's prefetch slots do not depend on the tile: compute them once
    int pl_y[NLOAD], pl_x[NLOAD];
#pragma unroll
    for (int k = 0; k < NLOAD; ++k) {
        const int idx = tid + k * NT;
        pl_y[k] = idx / RW;
        pl_x[k] = idx - pl_y[k] * RW;
    }
    float rx[NLOAD], ry[NLOAD];
    auto issue = [&](int tile) {
        const Tile tl = decode_tile<T2>(tile, tg, w, h);
        const size_t base = (size_t)tl.bz * (size_t)w * (size_t)h;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int gx = tl.x0 - RING - S + pl_x[k], gy = tl.y0 - RING - S + pl_y[k];
            const bool in = tid + k * NT < RH * RW && gx >= 0 && gx < w && gy >= 0 && gy < h;
            const size_t g = base + (size_t)gy * w + gx;
            rx[k] = in ? lx_in[g] : 0.0f;
            ry[k] = in ? ly_in[g] : 0.0f;
        }
    };
    // One H-pass position: window column wc (0..AW-1) of image column x, window row ly.
    auto h_pos = [&](int x, int wc, int ly, int x0, int y0) {
        const int y = y0 - RING - S + ly;
        if (x < 0 || x >= w || y < 0 || y >= h) return;
        const int cx = clampi(x, S, w - 1 - S), cy = clampi(y, S, h - 1 - S);
        const int o = (cy - (y0 - RING - S)) * RW + (cx - (x0 - RING - S));
        const float xa = sX[o - S], xb = sX[o], xc = sX[o + S];
        const float ya = sY[o - S], yc = sY[o + S];
        const int q = ly * AW + wc;
        sA[q] = ((0.0f + kn * xa) + kwn * xb) + kn * xc;
        sB[q] = (0.0f - ya) + yc;
        sC[q] = (0.0f - xa) + xc;
    };
    // One V-pass position: Ldet (+ second derivatives) at image (x, y); window column wc, window row ly.
    auto v_pos = [&](int x, int wc, int ly, int x0, int y0, size_t base, bool centre) {
        const int y = y0 - RING + ly;
        if (x < 0 || x >= w || y < 0 || y >= h) return;
        const int cx = clampi(x, S, w - 1 - S), cy = clampi(y, S, h - 1 - S);
        const int o = (cy - (y0 - RING - S)) * AW + (cx - (x0 - RING));
        const float lxx = (0.0f - sA[o - S * AW]) + sA[o + S * AW];
        const float lyy = ((0.0f + kn * sB[o - S * AW]) + kwn * sB[o]) + kn * sB[o + S * AW];
        const float lxy = ((0.0f + kn * sC[o - S * AW]) + kwn * sC[o]) + kn * sC[o + S * AW];
        const float det = ((lxx * lyy) - (lxy * lxy)) * quat;
        if (NMS) sD[ly * DW + wc] = det;
        if (centre) {
            const size_t g = base + (size_t)y * w + x;
            if (lxx_out) plane_store(lxx_out + g, lxx);
            if (lyy_out) plane_store(lyy_out + g, lyy);
            if (lxy_out) plane_store(lxy_out + g, lxy);
            plane_store(ldet_out + g, det);
        }
    };
    // The column passes of thread (tx, ty): rows ty, ty+8, ... of tile column x0+tx, fully unrolled and
    // split into "issue every LDS read" / "compute and store" so that the ~100-cycle ds_read latency of
    // one position overlaps the others (only 4 waves per SIMD are resident: LDS-limited occupancy).
    constexpr int HIT = (AH + NROWT - 1) / NROWT, VIT = (DH + NROWT - 1) / NROWT;
    auto h_column = [&](int x0, int y0) {
        const int x = x0 + tx;
        const bool xin = x < w;
        const int cxo = clampi(x, S, w - 1 - S) - (x0 - RING - S);
        float xa[HIT], xb[HIT], xc[HIT], ya[HIT], yc[HIT];
        bool ok[HIT];
#pragma unroll
        for (int i = 0; i < HIT; ++i) {
            const int ly = ty + i * NROWT, y = y0 - RING - S + ly;
            ok[i] = xin && ly < AH && y >= 0 && y < h;
            const int o = ok[i] ? (clampi(y, S, h - 1 - S) - (y0 - RING - S)) * RW + cxo : S;
            xa[i] = sX[o - S]; xb[i] = sX[o]; xc[i] = sX[o + S];
            ya[i] = sY[o - S]; yc[i] = sY[o + S];
        }
#pragma unroll
        for (int i = 0; i < HIT; ++i)
            if (ok[i]) {
                const int q = (ty + i * NROWT) * AW + tx + RING;
                sA[q] = ((0.0f + kn * xa[i]) + kwn * xb[i]) + kn * xc[i];
                sB[q] = (0.0f - ya[i]) + yc[i];
                sC[q] = (0.0f - xa[i]) + xc[i];
            }
    };
    auto v_column = [&](int x0, int y0, size_t base) {
        const int x = x0 + tx;
        const bool xin = x < w;
        const int cxo = clampi(x, S, w - 1 - S) - (x0 - RING);
        float a0[VIT], a2[VIT], b0[VIT], b1[VIT], b2[VIT], c0[VIT], c1[VIT], c2[VIT];
        bool ok[VIT];
#pragma unroll
        for (int i = 0; i < VIT; ++i) {
            const int ly = ty + i * NROWT, y = y0 - RING + ly;
            ok[i] = xin && ly < DH && y >= 0 && y < h;
            const int o = ok[i] ? (clampi(y, S, h - 1 - S) - (y0 - RING - S)) * AW + cxo : S * AW;
            a0[i] = sA[o - S * AW]; a2[i] = sA[o + S * AW];
            b0[i] = sB[o - S * AW]; b1[i] = sB[o]; b2[i] = sB[o + S * AW];
            c0[i] = sC[o - S * AW]; c1[i] = sC[o]; c2[i] = sC[o + S * AW];
        }
        float det[VIT];
#pragma unroll
        for (int i = 0; i < VIT; ++i) {
            const float lxx = (0.0f - a0[i]) + a2[i];
            const float lyy = ((0.0f + kn * b0[i]) + kwn * b1[i]) + kn * b2[i];
            const float lxy = ((0.0f + kn * c0[i]) + kwn * c1[i]) + kn * c2[i];
            det[i] = ((lxx * lyy) - (lxy * lxy)) * quat;
            const int ly = ty + i * NROWT;
            if (ok[i] && ly >= RING && ly < T2 + RING) {
                const size_t g = base + (size_t)(y0 - RING + ly) * w + x;
                if (lxx_out) plane_store(lxx_out + g, lxx);
                if (lyy_out) plane_store(lyy_out + g, lyy);
                if (lxy_out) plane_store(lxy_out + g, lxy);
                plane_store(ldet_out + g, det[i]);
            }
        }
        if (NMS) {  // sD aliases sX: every thread has finished reading sX before the barrier in front of this pass
#pragma unroll
            for (int i = 0; i < VIT; ++i)
                if (ok[i]) sD[(ty + i * NROWT) * DW + tx + RING] = det[i];
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * NT;
            if (idx < RH * RW) {
                sX[idx] = rx[k];
                sY[idx] = ry[k];
            }
        }
        __syncthreads();
        if (NMS && tid == 0) sCnt[0] = 0;
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        const Tile tl = decode_tile<T2>(tile, tg, w, h);
        const int x0 = tl.x0, y0 = tl.y0;
        const size_t base = (size_t)tl.bz * (size_t)w * (size_t)h;
        // ---- three H passes: thread (tx, ty) owns tile column x0+tx, window rows ty, ty+8, ... ----
        h_column(x0, y0);
        if (NMS && tid < 2 * AH) h_pos((tid & 1) ? x0 + TW : x0 - 1, (tid & 1) ? AW - 1 : 0, tid >> 1, x0, y0);
        __syncthreads();  // sX / sY are dead from here on
        // ---- three V passes + determinant ----
        v_column(x0, y0, base);
        if (NMS && tid < 2 * DH) v_pos((tid & 1) ? x0 + TW : x0 - 1, (tid & 1) ? DW - 1 : 0, tid >> 1, x0, y0, base, false);
        __syncthreads();
        if (NMS) {
            tile_extrema<DW, T2>(sD, sCnt, tid, tl, w, h, nms.thr, nms.border_m, nms.level, nms.cand, nms.cap, nms.count);
            __syncthreads();  // sD (= sX) is overwritten by the next iteration
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The whole detector of a level in one LDS-tiled kernel (detector_response.rs:8-55 + the extrema test):
// Lsmooth -> (H) Hm, Ho -> (V) Lx, Ly -> (H) A, B, C -> (V) Lxx, Lyy, Lxy, Ldet -> candidates.  Same window
// discipline as the kernels above (every window position holds the FILLED value of its stage), four passes
// with half width S, so the Lsmooth window is the tile grown by 2S (+1 for the extrema ring) on every side.
// Against k_deriv1 + k_deriv2 this reads Lsmooth once instead of Lsmooth + Lx + Ly (28 instead of 36 B/px
// with all planes kept) at the price of recomputing the first stage on the wider ring.  LDS: two buffers
// that are reused as the stages retire (Lsmooth -> Lx,Ly -> Ldet in one, Hm,Ho -> A,B,C in the other).
// ---------------------------------------------------------------------------------------------
// One launch covers a SET of levels that share sigma_size S (the kernel's template parameter): every level's
// detector depends only on its own Lsmooth, so the coarse levels of a pyramid — each too small to fill the chip —
// go into one grid (tiles numbered level after level).
struct DetLevel {
    const float* ls;
    float *lx, *ly, *lxx, *lyy, *lxy, *ldet;
    int w, h;
    TileGrid tg;
    int tile0;      // index of the level's first tile in the launch
    unsigned level;
    float border_m;
};
constexpr int kDetSetMax = 16;
struct DetSet {
    int nlevels, ntiles;
    float thr;
    Candidate* cand;
    unsigned cap;
    unsigned* count;
    DetLevel lv[kDetSetMax];
};

template <int S, bool NMS, bool KEEP>
__global__ void __launch_bounds__(NT)
k_detector_tiled(DetSet ds, float kn, float kwn, float quat) {
    constexpr int R = NMS ? 1 : 0;
    constexpr int DTH = det_tile_h(S);                                  // tile height (see det_tile_h)
    constexpr int W0W = TW + 2 * R + 4 * S, W0H = DTH + 2 * R + 4 * S;  // Lsmooth, origin (x0-R-2S, y0-R-2S)
    constexpr int H1W = TW + 2 * R + 2 * S, H1H = W0H;                 // Hm, Ho,   origin (x0-R-S,  y0-R-2S)
    constexpr int W2W = H1W, W2H = DTH + 2 * R + 2 * S;                // Lx, Ly,   origin (x0-R-S,  y0-R-S)
    constexpr int H2W = TW + 2 * R, H2H = W2H;                         // A, B, C,  origin (x0-R,    y0-R-S)
    constexpr int DW = H2W, DH = DTH + 2 * R;                          // Ldet,     origin (x0-R,    y0-R)
    constexpr int PSZ = (W0W * W0H > 2 * W2W * W2H) ? W0W * W0H : 2 * W2W * W2H;
    constexpr int QSZ = (2 * H1W * H1H > 3 * H2W * H2H) ? 2 * H1W * H1H : 3 * H2W * H2H;
    static_assert(DW * DH <= PSZ, "Ldet window must fit in the buffer it aliases");
    constexpr int NLOAD = (W0W * W0H + NT - 1) / NT;
    __shared__ float sP[PSZ];  // Lsmooth, then Lx | Ly, then Ldet
    __shared__ float sQ[QSZ];  // Hm | Ho, then A | B | C
    __shared__ unsigned sCnt[2];  // extrema of the tile, base slot in the candidate list
    float* const s0 = sP;
    float* const sHm = sQ;
    float* const sHo = sQ + H1W * H1H;
    float* const sLx = sP;
    float* const sLy = sP + W2W * W2H;
    float* const sA = sQ;
    float* const sB = sQ + H2W * H2H;
    float* const sC = sQ + 2 * H2W * H2H;
    float* const sD = sP;
    const int tid = threadIdx.x;
    const int ntiles = ds.ntiles;
    auto level_of = [&](int tile) {  // wave-uniform
        int li = 0;
        while (li + 1 < ds.nlevels && tile >= ds.lv[li + 1].tile0) ++li;
        return li;
    };
    float regs[NLOAD];
    auto issue = [&](int tile) {
        const DetLevel& dl = ds.lv[level_of(tile)];
        const int w = dl.w, h = dl.h;
        const Tile tl = decode_tile<DTH>(tile - dl.tile0, dl.tg, w, h);
        const float* src = dl.ls + (size_t)tl.bz * (size_t)w * (size_t)h;
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * NT;
            const int wy = idx / W0W, wx = idx - wy * W0W;
            const int gx = tl.x0 - R - 2 * S + wx, gy = tl.y0 - R - 2 * S + wy;
            regs[k] = (idx < W0W * W0H && gx >= 0 && gx < w && gy >= 0 && gy < h) ? src[(size_t)gy * w + gx] : 0.0f;
        }
    };
    int tile = blockIdx.x;
    if (tile < ntiles) issue(tile);
    for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) {
            const int idx = tid + k * NT;
            if (idx < W0W * W0H) s0[idx] = regs[k];
        }
        __syncthreads();
        if (NMS && tid == 0) sCnt[0] = 0;
        if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);
        const DetLevel& dl = ds.lv[level_of(tile)];
        const int w = dl.w, h = dl.h;
        const Tile tl = decode_tile<DTH>(tile - dl.tile0, dl.tg, w, h);
        const int x0 = tl.x0, y0 = tl.y0;
        const size_t base = (size_t)tl.bz * (size_t)w * (size_t)h;
        float* const lx_out = dl.lx;
        float* const ly_out = dl.ly;
        float* const lxx_out = dl.lxx;
        float* const lyy_out = dl.lyy;
        float* const lxy_out = dl.lxy;
        float* const ldet_out = dl.ldet;
        // A tile whose windows lie in the image's interior -- nine in ten of a 1080p level -- needs neither the in-image tests nor
        // the clamps of fill_border (every clamp is the identity there): the same arithmetic, two thirds of the instructions.
        const bool interior = x0 >= R + 3 * S && x0 + TW + R + 3 * S <= w && y0 >= R + 3 * S && y0 + DTH + R + 3 * S <= h;
        auto passes = [&](auto in_tag) {
            constexpr bool IN = decltype(in_tag)::value;
            // Every pass: a thread owns one window COLUMN and walks down it in strides of NT / width rows -- the column's
            // in-image test and clamp are formed once per pass, the row's cost an add and (off the interior) a clamp, where
            // an element index per trip cost a division by the window width and both clamps.
            // ---- stage 1, H pass: Hm = H_main(Ls), Ho = H_off(Ls) ----
            {
                constexpr int RS = NT / H1W;
                const int r0 = tid / H1W, wx = tid - r0 * H1W;
                const int x = x0 - R - S + wx;
                if (r0 < RS && (IN || (x >= 0 && x < w))) {
                    const int cx = IN ? x : clampi(x, S, w - 1 - S);
                    for (int wy = r0; wy < H1H; wy += RS) {
                        const int y = y0 - R - 2 * S + wy;
                        if (!IN && (y < 0 || y >= h)) continue;
                        const int cy = IN ? y : clampi(y, S, h - 1 - S);
                        const float* p = s0 + (cy - (y0 - R - 2 * S)) * W0W + (cx - (x0 - R - 2 * S));
                        const float a = p[-S], b = p[0], c = p[S];
                        sHm[wy * H1W + wx] = tap_main(a, b, c, kn, kwn);
                        sHo[wy * H1W + wx] = tap_off(a, b, c);
                    }
                }
            }
            __syncthreads();  // the Lsmooth window is dead from here on
            // ---- stage 1, V pass: Lx = V_off(Hm), Ly = V_main(Ho); the tile's own pixels go to HBM ----
            {
                constexpr int RS = NT / W2W;
                const int r0 = tid / W2W, wx = tid - r0 * W2W;
                const int x = x0 - R - S + wx;
                if (r0 < RS && (IN || (x >= 0 && x < w))) {
                    const int cx = IN ? x : clampi(x, S, w - 1 - S);
                    const bool own_x = x >= x0 && x < x0 + TW;
                    for (int wy = r0; wy < W2H; wy += RS) {
                        const int y = y0 - R - S + wy;
                        if (!IN && (y < 0 || y >= h)) continue;
                        const int cy = IN ? y : clampi(y, S, h - 1 - S);
                        const int o = (cy - (y0 - R - 2 * S)) * H1W + (cx - (x0 - R - S));
                        const float vx = tap_off(sHm[o - S * H1W], sHm[o], sHm[o + S * H1W]);
                        const float vy = tap_main(sHo[o - S * H1W], sHo[o], sHo[o + S * H1W], kn, kwn);
                        sLx[wy * W2W + wx] = vx;
                        sLy[wy * W2W + wx] = vy;
                        if (own_x && y >= y0 && y < y0 + DTH) {
                            const size_t gi = base + (size_t)y * w + x;
                            lx_out[gi] = vx;
                            ly_out[gi] = vy;
                        }
                    }
                }
            }
            __syncthreads();  // Hm / Ho are dead from here on
            // ---- stage 2, H pass: A = H_main(Lx), B = H_off(Ly), C = H_off(Lx) ----
            {
                constexpr int RS = NT / H2W;
                const int r0 = tid / H2W, wx = tid - r0 * H2W;
                const int x = x0 - R + wx;
                if (r0 < RS && (IN || (x >= 0 && x < w))) {
                    const int cx = IN ? x : clampi(x, S, w - 1 - S);
                    for (int wy = r0; wy < H2H; wy += RS) {
                        const int y = y0 - R - S + wy;
                        if (!IN && (y < 0 || y >= h)) continue;
                        const int cy = IN ? y : clampi(y, S, h - 1 - S);
                        const int o = (cy - (y0 - R - S)) * W2W + (cx - (x0 - R - S));
                        const float xa = sLx[o - S], xb = sLx[o], xc = sLx[o + S];
                        const float ya = sLy[o - S], yb = sLy[o], yc = sLy[o + S];
                        sA[wy * H2W + wx] = tap_main(xa, xb, xc, kn, kwn);
                        sB[wy * H2W + wx] = tap_off(ya, yb, yc);
                        sC[wy * H2W + wx] = tap_off(xa, xb, xc);
                    }
                }
            }
            __syncthreads();  // Lx / Ly windows are dead from here on (sD aliases them)
            // ---- stage 2, V pass + determinant ----
            {
                constexpr int RS = NT / DW;
                const int r0 = tid / DW, wx = tid - r0 * DW;
                const int x = x0 - R + wx;
                if (r0 < RS && (IN || (x >= 0 && x < w))) {
                    const int cx = IN ? x : clampi(x, S, w - 1 - S);
                    const bool own_x = x >= x0 && x < x0 + TW;
                    for (int wy = r0; wy < DH; wy += RS) {
                        const int y = y0 - R + wy;
                        if (!IN && (y < 0 || y >= h)) continue;
                        const int cy = IN ? y : clampi(y, S, h - 1 - S);
                        const int o = (cy - (y0 - R - S)) * H2W + (cx - (x0 - R));
                        const float lxx = tap_off(sA[o - S * H2W], sA[o], sA[o + S * H2W]);
                        const float lyy = tap_main(sB[o - S * H2W], sB[o], sB[o + S * H2W], kn, kwn);
                        const float lxy = tap_main(sC[o - S * H2W], sC[o], sC[o + S * H2W], kn, kwn);
                        const float det = ((lxx * lyy) - (lxy * lxy)) * quat;
                        if (NMS) sD[wy * DW + wx] = det;
                        if (own_x && y >= y0 && y < y0 + DTH) {
                            const size_t gi = base + (size_t)y * w + x;
                            if (KEEP) {
                                lxx_out[gi] = lxx;
                                lyy_out[gi] = lyy;
                                lxy_out[gi] = lxy;
                            }
                            ldet_out[gi] = det;
                        }
                    }
                }
            }
        };
        if (interior) passes(std::true_type{});
        else passes(std::false_type{});
        __syncthreads();
        if (NMS) {
            tile_extrema<DW, DTH>(sD, sCnt, tid, tl, w, h, ds.thr, dl.border_m, dl.level, ds.cand, ds.cap, ds.count);
            __syncthreads();  // sD (= sP) is overwritten by the next iteration
        }
    }
}

struct Launch {
    TileGrid tg;
    dim3 grid;
};
// workgroups of a persistent launch
inline long persist_blocks() { return (long)AKZ_PERSIST_BLOCKS; }
inline Launch plan_tiles(uint32_t w, uint32_t h, uint32_t n, int tile_h = TH) {
    Launch l;
    l.tg.tx = (int)((w + TW - 1) / TW);
    l.tg.ty = (int)((h + tile_h - 1) / tile_h);
    l.tg.n = (int)n;
    const long total = (long)l.tg.tx * l.tg.ty * l.tg.n;
    l.grid = dim3((unsigned)std::min<long>(total, persist_blocks()));
    return l;
}

}  // namespace

namespace launch {

bool blur_fused_supported(uint32_t ntaps) { return ntaps == 3 || ntaps == 5; }

template <typename T>
static void blur_fused_t(hipStream_t s, const T* in, float* out, uint32_t w, uint32_t h, uint32_t n, const float* k,
                         uint32_t ntaps) {
    DenseTaps t;
    for (uint32_t i = 0; i < (uint32_t)kMaxTaps; ++i) t.k[i] = i < ntaps ? k[i] : 0.0f;
    const Launch l = plan_tiles(w, h, n);
    if (ntaps == 3)
        hipLaunchKernelGGL((k_blur<1, T>), l.grid, dim3(NT), 0, s, in, out, (int)w, (int)h, l.tg, t);
    else
        hipLaunchKernelGGL((k_blur<2, T>), l.grid, dim3(NT), 0, s, in, out, (int)w, (int)h, l.tg, t);
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
    const Launch l = plan_tiles(w, h, n);
    if (half)
        hipLaunchKernelGGL((k_prep<true>), l.grid, dim3(NT), 0, s, prev, lt_out, lsmooth, lflow, (int)w, (int)h,
                           (int)pw, (int)ph, l.tg, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow);
    else
        hipLaunchKernelGGL((k_prep<false>), l.grid, dim3(NT), 0, s, prev, lt_out, lsmooth, lflow, (int)w, (int)h,
                           (int)pw, (int)ph, l.tg, g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1], d_k, k_pow);
}

bool head_fused_supported(uint32_t w, uint32_t h, uint32_t ntaps0, uint32_t ntaps1) { return ntaps0 == 5 && ntaps1 == 3 && w >= 5 && h >= 5; }
// Level 0 of a small job: Lt0, the contrast factor's blurred image and the largest squared gradient magnitude per image (k_head);
// d_smax_bits must be zero before (the histogram pass leaves it so)
template <typename T>
static void head_fused_t(hipStream_t s, const T* in, float* lt0, float* blurred, float* gx, float* gy, uint32_t w, uint32_t h, uint32_t n,
                         const float* k5, const float* g3, unsigned long long* d_smax_bits, uint32_t* d_zero_word) {
    DenseTaps t;
    for (uint32_t i = 0; i < (uint32_t)kMaxTaps; ++i) t.k[i] = i < 5 ? k5[i] : 0.0f;
    const Taps m = taps_scharr_main(1);
    const Launch l = plan_tiles(w, h, n);
    hipLaunchKernelGGL((k_head<T>), l.grid, dim3(NT), 0, s, in, lt0, blurred, gx, gy, (int)w, (int)h, l.tg, t,
                       PrepTaps{g3[0], g3[1], g3[2], m.wgt[0], m.wgt[1]}, d_smax_bits, d_zero_word);
}
void head_fused_u8(hipStream_t s, const uint8_t* in, float* lt0, float* blurred, float* gx, float* gy, uint32_t w, uint32_t h, uint32_t n,
                   const float* k5, const float* g3, unsigned long long* d_smax_bits, uint32_t* d_zero_word) {
    head_fused_t<uint8_t>(s, in, lt0, blurred, gx, gy, w, h, n, k5, g3, d_smax_bits, d_zero_word);
}
void head_fused_f32(hipStream_t s, const float* in, float* lt0, float* blurred, float* gx, float* gy, uint32_t w, uint32_t h, uint32_t n,
                    const float* k5, const float* g3, unsigned long long* d_smax_bits, uint32_t* d_zero_word) {
    head_fused_t<float>(s, in, lt0, blurred, gx, gy, w, h, n, k5, g3, d_smax_bits, d_zero_word);
}

bool detector_fused_supported(uint32_t sigma) { return sigma >= 1 && sigma <= 6; }
bool detector_nms_fused_supported(uint32_t sigma) { return sigma >= 1 && sigma <= 4; }  // LDS <= 64 KB

#define AKZ_DET(S, NMSF)                                                                                           \
    case S:                                                                                                        \
        hipLaunchKernelGGL((k_deriv1<S>), l.grid, dim3(NT), 0, s, lsmooth, lx, ly, (int)w, (int)h, l.tg, kn, kwn); \
        hipLaunchKernelGGL((k_deriv2<S, NMSF>), l2.grid, dim3(NT), 0, s, (const float*)lx, (const float*)ly, lxx,  \
                           lyy, lxy, ldet_out, (int)w, (int)h, l2.tg, kn, kwn, quat, na);                          \
        break;

void detector_fused(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx, float* lyy,
                    float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n) {
    const Taps m = taps_scharr_main(sigma);
    const float kn = m.wgt[0], kwn = m.wgt[1];
    const float quat = (float)(sigma * sigma * sigma * sigma);
    const Launch l = plan_tiles(w, h, n), l2 = plan_tiles(w, h, n, deriv2_tile_h((int)sigma));
    const NmsArgs na{};
    switch (sigma) {
        AKZ_DET(1, false) AKZ_DET(2, false) AKZ_DET(3, false) AKZ_DET(4, false) AKZ_DET(5, false) AKZ_DET(6, false)
        default: break;
    }
}

void detector_nms_fused(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx,
                        float* lyy, float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n, uint32_t level,
                        float thr, float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count) {
    const Taps m = taps_scharr_main(sigma);
    const float kn = m.wgt[0], kwn = m.wgt[1];
    const float quat = (float)(sigma * sigma * sigma * sigma);
    const Launch l = plan_tiles(w, h, n), l2 = plan_tiles(w, h, n, deriv2_tile_h((int)sigma));
    const NmsArgs na{level, thr, border_m, d_cand, cap, d_count};
    switch (sigma) {
        AKZ_DET(1, true) AKZ_DET(2, true) AKZ_DET(3, true) AKZ_DET(4, true)
        default: break;
    }
}
#undef AKZ_DET

// One-kernel LDS-tiled detector (k_detector_tiled) over a set of levels with the same sigma_size (<= 4);
// d_cand == nullptr: no extrema test.  lxx/lyy/lxy of a level may be null together (planes not kept): `keep`
// must then be false for the whole set.
bool detector_tiled_fused_supported(uint32_t sigma) { return sigma >= 1 && sigma <= 4; }
#define AKZ_TDET(S)                                                                                                   \
    case S:                                                                                                           \
        if (d_cand && keep)                                                                                           \
            hipLaunchKernelGGL((k_detector_tiled<S, true, true>), grid, dim3(NT), 0, s, ds, kn, kwn, quat);           \
        else if (d_cand)                                                                                              \
            hipLaunchKernelGGL((k_detector_tiled<S, true, false>), grid, dim3(NT), 0, s, ds, kn, kwn, quat);          \
        else if (keep)                                                                                                \
            hipLaunchKernelGGL((k_detector_tiled<S, false, true>), grid, dim3(NT), 0, s, ds, kn, kwn, quat);          \
        else                                                                                                          \
            hipLaunchKernelGGL((k_detector_tiled<S, false, false>), grid, dim3(NT), 0, s, ds, kn, kwn, quat);         \
        break;
uint32_t detector_tiled_set_max() { return (uint32_t)kDetSetMax; }
void detector_tiled_set(hipStream_t s, uint32_t sigma, const DetLevelDesc* levels, uint32_t nlevels, uint32_t n, float thr,
                        Candidate* d_cand, uint32_t cap, uint32_t* d_count) {
    if (nlevels == 0 || nlevels > (uint32_t)kDetSetMax) return;
    const Taps m = taps_scharr_main(sigma);
    const float kn = m.wgt[0], kwn = m.wgt[1];
    const float quat = (float)(sigma * sigma * sigma * sigma);
    DetSet ds;
    ds.nlevels = (int)nlevels;
    ds.thr = thr;
    ds.cand = d_cand;
    ds.cap = cap;
    ds.count = d_count;
    bool keep = true;
    long total = 0;
    for (uint32_t i = 0; i < nlevels; ++i) {
        const DetLevelDesc& d = levels[i];
        DetLevel& o = ds.lv[i];
        o.ls = d.lsmooth; o.lx = d.lx; o.ly = d.ly; o.lxx = d.lxx; o.lyy = d.lyy; o.lxy = d.lxy; o.ldet = d.ldet;
        o.w = (int)d.w; o.h = (int)d.h;
        o.tg = plan_tiles(d.w, d.h, n, det_tile_h((int)sigma)).tg;
        o.tile0 = (int)total;
        o.level = d.level;
        o.border_m = d.border_m;
        total += (long)o.tg.tx * o.tg.ty * o.tg.n;
        keep = keep && d.lxx && d.lyy && d.lxy;
    }
    ds.ntiles = (int)total;
    const dim3 grid((unsigned)std::min<long>(total, persist_blocks()));
    switch (sigma) {
        AKZ_TDET(1) AKZ_TDET(2) AKZ_TDET(3) AKZ_TDET(4)
        default: break;
    }
}
void detector_tiled_fused(hipStream_t s, const float* lsmooth, uint32_t sigma, float* lx, float* ly, float* lxx,
                          float* lyy, float* lxy, float* ldet_out, uint32_t w, uint32_t h, uint32_t n, uint32_t level,
                          float thr, float border_m, Candidate* d_cand, uint32_t cap, uint32_t* d_count) {
    const DetLevelDesc d{lsmooth, lx, ly, lxx, lyy, lxy, ldet_out, w, h, level, border_m};
    detector_tiled_set(s, sigma, &d, 1, n, thr, d_cand, cap, d_count);
}
#undef AKZ_TDET

}  // namespace launch
}  // namespace akz
