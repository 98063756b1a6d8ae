// pm_g2 of one pixel (akaze/src/lib.rs:30-37): g = (1.0 / (1.0 + inverse_k * (lx*lx + ly*ly))) as f32, all in f64.
// Device code only; shared by every kernel family so that there is one statement of it.
//
// The f64 value is only ever seen after its rounding to f32, so the correctly rounded f64 quotient (v_div_scale x2, v_rcp,
// 4 fma, mul, fma, v_div_fmas, v_div_fixup: eleven f64 instructions) is needed only where it decides that rounding:
//   * r = v_rcp_f64(x) refined by two Newton steps is within 1 ulp(f64) of the correctly rounded 1/x (the hardware
//     reciprocal is good to ~2^-23; each step squares the error; the last fma rounds once);
//   * the f32 rounding of two doubles that close differs only if an f32 rounding boundary -- a double whose low 29
//     mantissa bits are exactly 0x10000000 -- lies between them.  So: if r's low 29 bits are further than 8 ulp from that
//     pattern, (float)r == (float)(1.0 / x) and r is used; otherwise (17 of 2^29 doubles: one pixel in 30 million) the lane
//     redoes the quotient with the full division.  x >= 2^64 (f32-denormal quotients round at another bit), infinities
//     and NaNs take the division as well.
// tests/test_gpu_ops.py::test_pm_g2_reciprocal_boundaries drives both paths on doubles placed on and around the boundaries.
#pragma once
#include <hip/hip_runtime.h>

namespace akz {

__device__ __forceinline__ float rcp_f64_to_f32(double x) {  // (1.0 / x) as f32 for x >= 1.0 (anything else: the division)
    const double r0 = __builtin_amdgcn_rcp(x);
    const double e0 = __builtin_fma(-x, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-x, r1, 1.0);
    double r = __builtin_fma(r1, e1, r1);
    const unsigned lo = (unsigned)__double_as_longlong(r);
    const unsigned hi = (unsigned)(__double_as_longlong(x) >> 32);
    const bool near_boundary = ((lo + (8u - 0x10000000u)) & 0x1FFFFFFFu) <= 16u;
    const bool plain = hi >= 0x3FF00000u && hi < 0x43F00000u;  // 1.0 <= x < 2^64, not NaN
    if (__builtin_expect(near_boundary || !plain, 0)) r = 1.0 / x;
    return (float)r;
}

__device__ __forceinline__ float pm_g2_px(float lx, float ly, double inverse_k) {  // lib.rs:30-37
    const double dx = (double)lx, dy = (double)ly;
    return rcp_f64_to_f32(1.0 + inverse_k * (dx * dx + dy * dy));
}

}  // namespace akz
