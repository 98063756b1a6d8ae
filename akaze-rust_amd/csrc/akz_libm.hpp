// atan2f / sinf / cosf as THIS MACHINE'S libm computes them, for the device (and, compiled for the host, for the CPU tests).
//
// The reference takes the main orientation with f32::atan2 and rotates the M-LDB pattern with f32::cos / f32::sin
// (scale_space_extrema.rs:326, descriptors.rs:55-56): Rust's std calls the platform libm, so "the reference's result" is
// what glibc's atan2f / cosf / sinf return on the machine the reference runs on.  The library used to fetch every
// keypoint's orientation sums to the host for those three calls -- one host round trip in the middle of the finish half, a
// tenth of a lone frame's latency.  These are the same functions as straight-line IEEE arithmetic the GPU can run:
//
//   atan2f, atanf   glibc 2.35 sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c (the fdlibm float code: f32 adds, multiplies and
//                   divisions in source order; no multiarch variant on x86-64)
//   sinf, cosf      glibc 2.35 sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h (f64 polynomials on a reduced argument).
//                   x86-64 glibc selects one of two builds at load time: the FMA build (every `a + b * c` of the source is one
//                   fused multiply-add -- read off the disassembly of libm.so.6) and the SSE2 build (never fused).
//                   `FMA` picks between them.
//
// Nothing here is trusted blindly: a context that wants to use these runs BOTH against the host's libm on a few million
// arguments first (akz_extract.cpp: device_libm_mode) and keeps the host round trip unless one variant reproduces every
// bit; tests/test_libm.py holds the host build of this header to libm on dense argument sets, tests/test_gpu_libm.py the
// device build.  Arguments outside the supported range (|x| >= 120 for sinf / cosf: an orientation is never outside
// [-pi, pi]) return NaN and raise `*unsupported`, which sends the job back to the host's libm.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AKZ_LIBM_FN __host__ __device__ __forceinline__
#else
#define AKZ_LIBM_FN inline
#endif

namespace akz {
namespace libm {

AKZ_LIBM_FN uint32_t f2u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}
AKZ_LIBM_FN float u2f(uint32_t u) {
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// ---- s_atanf.c ----
AKZ_LIBM_FN float atanf_glibc(float x) {
    const float atanhi[4] = {u2f(0x3eed6338u), u2f(0x3f490fdau), u2f(0x3f7b985eu), u2f(0x3fc90fdau)};
    const float atanlo[4] = {u2f(0x31ac3769u), u2f(0x33222168u), u2f(0x33140fb4u), u2f(0x33a22168u)};
    const float aT[11] = {u2f(0x3eaaaaabu), u2f(0xbe4ccccdu), u2f(0x3e124925u), u2f(0xbde38e38u), u2f(0x3dba2e6eu), u2f(0xbd9d8795u),
                          u2f(0x3d886b35u), u2f(0xbd6ef16bu), u2f(0x3d4bda59u), u2f(0xbd15a221u), u2f(0x3c8569d7u)};
    const float one = 1.0f;
    const int32_t hx = (int32_t)f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;  // NaN
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {      // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29 (huge + x > one)
        id = -1;
    } else {
        x = u2f((uint32_t)ix);  // fabsf
        if (ix < 0x3f980000) {  // |x| < 1.1875
            if (ix < 0x3f300000) {  // 7/16 <= |x| < 11/16
                id = 0;
                x = (2.0f * x - one) / (2.0f + x);
            } else {  // 11/16 <= |x| < 19/16
                id = 1;
                x = (x - one) / (x + one);
            }
        } else {
            if (ix < 0x401c0000) {  // |x| < 2.4375
                id = 2;
                x = (x - 1.5f) / (one + 1.5f * x);
            } else {  // 2.4375 <= |x| < 2^25
                id = 3;
                x = -1.0f / x;
            }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    const float r = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -r : r;
}

// ---- e_atan2f.c ----
AKZ_LIBM_FN float atan2f_glibc(float y, float x) {
    const float tiny = u2f(0x0da24260u);       // 1.0e-30
    const float pi_o_4 = u2f(0x3f490fdbu), pi_o_2 = u2f(0x3fc90fdbu), pi = u2f(0x40490fdbu), pi_lo = u2f(0xb3bbbd2eu);
    const int32_t hx = (int32_t)f2u(x), hy = (int32_t)f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;  // NaN
    if (hx == 0x3f800000) return atanf_glibc(y);           // x = 1.0
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);  // 2 * sign(x) + sign(y)
    if (iy == 0) {  // y = 0
        switch (m) {
            case 0:
            case 1: return y;
            case 2: return pi + tiny;
            default: return -pi - tiny;
        }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // x = 0
    if (ix == 0x7f800000) {  // x is INF
        if (iy == 0x7f800000) {
            switch (m) {
                case 0: return pi_o_4 + tiny;
                case 1: return -pi_o_4 - tiny;
                case 2: return 3.0f * pi_o_4 + tiny;
                default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
                case 0: return 0.0f;
                case 1: return -0.0f;
                case 2: return pi + tiny;
                default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;  // y is INF
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;      // |y / x| > 2^60
    else if (hx < 0 && k < -60) z = 0.0f;        // |y| / x < -2^60
    else z = atanf_glibc(u2f(f2u(y / x) & 0x7fffffffu));  // atanf(fabsf(y / x))
    switch (m) {
        case 0: return z;
        case 1: return u2f(f2u(z) ^ 0x80000000u);
        case 2: return pi - (z - pi_lo);
        default: return (z - pi_lo) - pi;
    }
}

// ---- sincosf.h: the two polynomials on the reduced argument (double), `a + b * c` fused or not ----
template <bool FMA>
AKZ_LIBM_FN double mad(double b, double c, double a) {  // a + b * c
    if (FMA) return __builtin_fma(b, c, a);
    const double p = b * c;  // (separate statements and -ffp-contract=off: never fused)
    return a + p;
}
struct SinCosTable {
    double c0, c1, c2, c3, c4, s1, s2, s3;
};
AKZ_LIBM_FN SinCosTable sincos_table(bool negated) {  // __sincosf_table[0] / [1] (the second: cos coefficients negated)
    SinCosTable t;
    t.c0 = 0x1p0, t.c1 = -0x1.ffffffd0c621cp-2, t.c2 = 0x1.55553e1068f19p-5, t.c3 = -0x1.6c087e89a359dp-10, t.c4 = 0x1.99343027bf8c3p-16;
    t.s1 = -0x1.555545995a603p-3, t.s2 = 0x1.1107605230bc4p-7, t.s3 = -0x1.994eb3774cf24p-13;
    if (negated) t.c0 = -t.c0, t.c1 = -t.c1, t.c2 = -t.c2, t.c3 = -t.c3, t.c4 = -t.c4;
    return t;
}
// sinf_poly (sincosf.h): the sine polynomial for even n, the cosine polynomial for odd n
template <bool FMA>
AKZ_LIBM_FN float sinf_poly(double x, double x2, const SinCosTable& p, int n) {
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = mad<FMA>(x2, p.s3, p.s2);
        const double x7 = x3 * x2;
        const double s = mad<FMA>(x3, p.s1, x);
        return (float)mad<FMA>(x7, s1, s);
    }
    const double x4 = x2 * x2;
    const double c2 = mad<FMA>(x2, p.c4, p.c3);
    const double c1 = mad<FMA>(x2, p.c1, p.c0);
    const double x6 = x4 * x2;
    const double c = mad<FMA>(x4, p.c2, c1);
    return (float)mad<FMA>(x6, c2, c);
}
AKZ_LIBM_FN uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ffu; }
// reduce_fast (sincosf.h): x - n * pi/2 for |x| < 120, n the nearest integer to x * 2/pi
template <bool FMA>
AKZ_LIBM_FN double reduce_fast(double x, int* np) {
    const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return mad<FMA>(-(double)n, hpi, x);  // x - n * hpi (vfnmadd in the FMA build: the product's sign is exact either way)
}
// sinf / cosf for |y| < 120; beyond (and for infinities and NaNs) NaN with *unsupported raised
template <bool FMA, bool COS>
AKZ_LIBM_FN float sincosf_glibc(float y, bool* unsupported) {
    const double sign[4] = {1.0, -1.0, -1.0, 1.0};
    double x = (double)y;
    const uint32_t top = abstop12(y);
    if (top < 0x3f4u) {  // |y| < pi/4 (abstop12(pio4f))
        const double x2 = x * x;
        if (top < 0x398u) return COS ? 1.0f : y;  // |y| < 2^-12
        return sinf_poly<FMA>(x, x2, sincos_table(false), COS ? 1 : 0);
    }
    if (top < 0x42fu) {  // |y| < 120
        int n;
        x = reduce_fast<FMA>(x, &n);
        const double s = sign[n & 3];
        const SinCosTable p = sincos_table((n & 2) != 0);
        return sinf_poly<FMA>(x * s, x * x, p, COS ? n ^ 1 : n);
    }
    *unsupported = true;
    return u2f(0x7fc00000u);
}
template <bool FMA>
AKZ_LIBM_FN float sinf_glibc(float y, bool* unsupported) { return sincosf_glibc<FMA, false>(y, unsupported); }
template <bool FMA>
AKZ_LIBM_FN float cosf_glibc(float y, bool* unsupported) { return sincosf_glibc<FMA, true>(y, unsupported); }

}  // namespace libm
}  // namespace akz
