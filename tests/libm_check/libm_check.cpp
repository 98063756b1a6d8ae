// Test infrastructure: csrc/akz_libm.hpp compiled for the HOST next to this machine's libm, array by array
// (tests/test_libm.py).  which: 0 = libm, 1 = the header's FMA build, 2 = its SSE2 build.
#include <cmath>
#include <cstdint>

#include "../../akaze-rust_amd/csrc/akz_libm.hpp"

extern "C" {
void lc_atan2f(int which, const float* y, const float* x, float* out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) out[i] = which ? akz::libm::atan2f_glibc(y[i], x[i]) : atan2f(y[i], x[i]);
}
// *unsupported: arguments the header refuses (|x| >= 120, infinities, NaNs)
void lc_sinf(int which, const float* x, float* out, uint64_t n, uint64_t* unsupported) {
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        bool u = false;
        out[i] = which == 0 ? sinf(x[i]) : which == 1 ? akz::libm::sinf_glibc<true>(x[i], &u) : akz::libm::sinf_glibc<false>(x[i], &u);
        bad += u;
    }
    if (unsupported) *unsupported = bad;
}
void lc_cosf(int which, const float* x, float* out, uint64_t n, uint64_t* unsupported) {
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i) {
        bool u = false;
        out[i] = which == 0 ? cosf(x[i]) : which == 1 ? akz::libm::cosf_glibc<true>(x[i], &u) : akz::libm::cosf_glibc<false>(x[i], &u);
        bad += u;
    }
    if (unsupported) *unsupported = bad;
}
}
