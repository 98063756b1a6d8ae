// One level's preparation on a tile whose input window already lies in LDS (lib.rs:92-105): Lt_i (the previous level's final
// Lt) -> Lsmooth_i = gaussian_blur(Lt_i, 1.0) -> scale-1 Scharr pair -> Lflow_i = pm_g2.  Four passes of half width 1, each
// with the reference's fill_border (the raw result at the coordinates clamped to the interior, types/image.rs:239-332);
// windows: input +-2, H_g rows +-2 cols +-1, Lsmooth +-1, H_scharr rows +-1.
//
// Shared by k_prep (akz_stencil.hip: the window is loaded from the previous level's plane) and by k_fed_own's epilogue
// (akz_kernels.hip: the window is what the level's last diffusion launch has just computed -- the next level's preparation
// without a launch of its own), so that there is one statement of the arithmetic.
//
// sI: (TH + 4) x (TW + 4) input window, origin (x0 - 2, y0 - 2); pixels outside the image are never read.
// sA: (TH + 4) x (TW + 2) scratch; sB: (TH + 2) x (TW + 2) scratch.  The Scharr H-pass windows reuse sI and sA (dead by then).
// All threads of the workgroup call it together (it synchronises); the caller synchronises before (sI complete) and, if it
// goes on using the buffers, after.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "akz_pm_g2.hpp"

namespace akz {

struct PrepTaps {
    float g0, g1, g2;  // gaussian_kernel(1.0, 3)
    float kn, kwn;     // scharr main-axis taps [kn, kwn, kn] at scale 1
};

// `fin(x, y, lx, ly)`: what becomes of the Scharr pair of pixel (x, y) -- evaluated, like every stage, at the coordinates
// clamped to the interior: the preparation stores pm_g2 of it, the level-0 kernel (k_head) folds it into the contrast
// factor's maximum.
template <int TW, int TH, int NT, class Fin>
__device__ __forceinline__ void prep_passes_fin(float* __restrict__ sI, float* __restrict__ sA, float* __restrict__ sB, int x0, int y0, int w,
                                                int h, size_t base, float* __restrict__ lsmooth, PrepTaps t, Fin fin) {
    constexpr int IW = TW + 4;
    constexpr int AW = TW + 2, AH = TH + 4;
    constexpr int BW = TW + 2, BH = TH + 2;
    constexpr int CH = TH + 2;
    static_assert(CH * TW <= (TH + 4) * IW && CH * TW <= AH * AW, "the Scharr windows fit the windows they alias");
    float* const sM = sI;
    float* const sO = sA;
    const int tid = threadIdx.x;
    auto clampi = [](int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); };
    // a tile whose windows lie in the image's interior needs neither the in-image tests nor fill_border's clamps (all identities)
    const bool interior = x0 >= 3 && x0 + TW + 3 <= w && y0 >= 3 && y0 + TH + 3 <= h;
    auto passes = [&](auto in_tag) {
        constexpr bool IN = decltype(in_tag)::value;
        for (int idx = tid; idx < AH * AW; idx += NT) {  // A = H_g(in)
            const int ly = idx / AW, lx = idx - ly * AW;
            const int x = x0 - 1 + lx, y = y0 - 2 + ly;
            if (IN || (x >= 0 && x < w && y >= 0 && y < h)) {
                const int cx = IN ? x : clampi(x, 1, w - 2), cy = IN ? y : clampi(y, 1, h - 2);
                const float* p = sI + (cy - (y0 - 2)) * IW + (cx - (x0 - 2));
                sA[idx] = ((0.0f + t.g0 * p[-1]) + t.g1 * p[0]) + t.g2 * p[1];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < BH * BW; idx += NT) {  // B = Lsmooth = V_g(A)
            const int ly = idx / BW, lx = idx - ly * BW;
            const int x = x0 - 1 + lx, y = y0 - 1 + ly;
            if (IN || (x >= 0 && x < w && y >= 0 && y < h)) {
                const int cx = IN ? x : clampi(x, 1, w - 2), cy = IN ? y : clampi(y, 1, h - 2);
                const float* p = sA + (cy - (y0 - 2)) * AW + (cx - (x0 - 1));
                const float v = ((0.0f + t.g0 * p[-AW]) + t.g1 * p[0]) + t.g2 * p[AW];
                sB[idx] = v;
                if (lx >= 1 && lx <= TW && ly >= 1 && ly <= TH) lsmooth[base + (size_t)y * w + x] = v;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < CH * TW; idx += NT) {  // H passes of the Scharr pair (derivatives.rs:41-65)
            const int ly = idx / TW, lx = idx - ly * TW;
            const int x = x0 + lx, y = y0 - 1 + ly;
            if (IN || (x < w && y >= 0 && y < h)) {
                const int cx = IN ? x : clampi(x, 1, w - 2), cy = IN ? y : clampi(y, 1, h - 2);
                const float* p = sB + (cy - (y0 - 1)) * BW + (cx - (x0 - 1));
                const float a = p[-1], b = p[0], c = p[1];
                sM[idx] = ((0.0f + t.kn * a) + t.kwn * b) + t.kn * c;
                sO[idx] = (0.0f - a) + c;
            }
        }
        __syncthreads();
        for (int idx = tid; idx < TH * TW; idx += NT) {  // V passes + pm_g2
            const int ly = idx / TW, lx = idx - ly * TW;
            const int x = x0 + lx, y = y0 + ly;
            if (IN || (x < w && y < h)) {
                const int cx = IN ? x : clampi(x, 1, w - 2), cy = IN ? y : clampi(y, 1, h - 2);
                const int o = (cy - (y0 - 1)) * TW + (cx - x0);
                const float lx1 = (0.0f - sM[o - TW]) + sM[o + TW];
                const float ly1 = ((0.0f + t.kn * sO[o - TW]) + t.kwn * sO[o]) + t.kn * sO[o + TW];
                fin(x, y, lx1, ly1);
            }
        }
    };
    if (interior) passes(std::true_type{});
    else passes(std::false_type{});
}
template <int TW, int TH, int NT>
__device__ __forceinline__ void prep_passes(float* __restrict__ sI, float* __restrict__ sA, float* __restrict__ sB, int x0, int y0, int w,
                                            int h, size_t base, float* __restrict__ lsmooth, float* __restrict__ lflow, PrepTaps t,
                                            double inverse_k) {
    prep_passes_fin<TW, TH, NT>(sI, sA, sB, x0, y0, w, h, base, lsmooth, t, [&](int x, int y, float lx1, float ly1) {
        lflow[base + (size_t)y * w + x] = pm_g2_px(lx1, ly1, inverse_k);
    });
}

}  // namespace akz
