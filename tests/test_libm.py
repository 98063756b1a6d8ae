"""csrc/akz_libm.hpp -- glibc's atan2f / cosf / sinf restated as IEEE arithmetic so that the GPU can form a keypoint's angle and
its cosine / sine without a host round trip -- compiled for the HOST and held to this machine's libm, bit for bit.  The
device build of the same header is held to libm by tests/test_gpu_libm.py and, in every process that uses it, by the
library's own self-test (akz_extract.cpp: device_libm_mode)."""
import ctypes as C
import os
import platform
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.skipif(platform.machine() != "x86_64" or platform.libc_ver()[0] != "glibc",
                                reason="the header restates x86-64 glibc's float routines")


@pytest.fixture(scope="module")
def L():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "libm_check")], stdout=subprocess.DEVNULL)
    return C.CDLL(os.path.join(HERE, "libm_check", "liblibm_check.so"))


def fp(a):
    return a.ctypes.data_as(C.c_void_p)


def host_build():
    """the build of sinf / cosf glibc's selector takes on this CPU: 1 = FMA, 2 = SSE2 (0: FMA4, not restated)"""
    flags = open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0].split()
    if "fma" in flags and "avx2" in flags:
        return 1
    return 0 if "fma4" in flags else 2


def test_sinf_cosf_match_libm_on_every_eleventh_float_below_120(L):
    which = host_build()
    if which == 0:
        pytest.skip("glibc runs its FMA4 build here")
    pos = np.arange(0x39000000, 0x42f00000, 11, dtype=np.uint32)   # [2^-13, 120)
    small = np.arange(0, 0x39000000, 4099, dtype=np.uint32)        # zero, subnormals, everything below 2^-13
    x = np.concatenate([pos, pos | 0x80000000, small, small | 0x80000000]).view(np.float32)
    for fn in (L.lc_sinf, L.lc_cosf):
        ref, got = np.empty_like(x), np.empty_like(x)
        bad = C.c_uint64()
        fn(0, fp(x), fp(ref), C.c_uint64(len(x)), None)
        fn(which, fp(x), fp(got), C.c_uint64(len(x)), C.byref(bad))
        assert bad.value == 0
        diff = np.nonzero(ref.view(np.uint32) != got.view(np.uint32))[0]
        assert len(diff) == 0, (x[diff[:5]], ref[diff[:5]], got[diff[:5]])
    # arguments the header does not cover are refused, never answered wrongly
    far = np.array([120.0, -1e9, np.inf, -np.inf, np.nan], np.float32)
    out, bad = np.empty_like(far), C.c_uint64()
    L.lc_sinf(which, fp(far), fp(out), C.c_uint64(len(far)), C.byref(bad))
    assert bad.value == len(far) and np.isnan(out).all()


def test_the_two_builds_of_sinf_cosf_differ_somewhere(L):
    """Why the build matters: glibc's FMA and SSE2 builds of sinf / cosf disagree on a few dozen of the 3.3 x 10^8 floats below
    120 (an exhaustive run of this library found 12 + 22) -- all of them beyond |x| = 17, none inside [-pi, pi] where an
    orientation lies.  Some of them here; on each the header's matching build equals libm, the other one does not."""
    which = host_build()
    if which == 0:
        pytest.skip("glibc runs its FMA4 build here")
    for fn, bits in ((L.lc_sinf, [0x4255b0a9, 0x42a35c07, 0x42a35d44]), (L.lc_cosf, [0x418a3adb, 0x418a3adc, 0x4202eb4b, 0x4280ce28])):
        x = np.array(bits, np.uint32).view(np.float32)
        ref, fma, sse = np.empty_like(x), np.empty_like(x), np.empty_like(x)
        for w, o in ((0, ref), (1, fma), (2, sse)):
            fn(w, fp(x), fp(o), C.c_uint64(len(x)), None)
        assert (fma.view(np.uint32) != sse.view(np.uint32)).all()
        mine, other = (fma, sse) if which == 1 else (sse, fma)
        assert (mine.view(np.uint32) == ref.view(np.uint32)).all() and (other.view(np.uint32) != ref.view(np.uint32)).all()
    # ... and on [-pi, pi] (and a little beyond) the two builds are the same function (every float in the exhaustive run; every
    # fifth here, for the suite's time)
    pos = np.arange(0, 0x40800000, 5, dtype=np.uint32)   # [0, 4)
    for lo in range(0, len(pos), 1 << 26):
        x = pos[lo:lo + (1 << 26)].view(np.float32)
        for fn in (L.lc_sinf, L.lc_cosf):
            a, b = np.empty_like(x), np.empty_like(x)
            fn(1, fp(x), fp(a), C.c_uint64(len(x)), None)
            fn(2, fp(x), fp(b), C.c_uint64(len(x)), None)
            assert (a.view(np.uint32) == b.view(np.uint32)).all()


def test_atan2f_matches_libm(L):
    rng = np.random.default_rng(3)
    n = 6_000_000

    def mag(lo, hi):
        return (2.0 ** rng.uniform(lo, hi, n)).astype(np.float32) * rng.choice(np.float32([-1, 1]), n)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 2.0 ** -126, 2.0 ** -149, 3.4028235e38, 0.4375, 0.6875, 1.1875, 2.4375,
                   2.0 ** 25, 2.0 ** -29, 2.0 ** 61, 2.0 ** -61], np.float32)
    cases = [(mag(-100, 100), mag(-100, 100)), (mag(-3, 3), mag(-3, 3)), (mag(-30, 30), np.ones(n, np.float32)),
             (np.repeat(sp, len(sp)), np.tile(sp, len(sp)))]
    y0 = mag(-20, 20)
    cases.append((y0, (y0 * rng.uniform(0.3, 3.0, n).astype(np.float32))))   # ratios around every breakpoint of atanf's reduction
    for y, x in cases:
        y, x = np.ascontiguousarray(y), np.ascontiguousarray(x)
        ref, got = np.empty_like(y), np.empty_like(y)
        L.lc_atan2f(0, fp(y), fp(x), fp(ref), C.c_uint64(len(y)))
        L.lc_atan2f(1, fp(y), fp(x), fp(got), C.c_uint64(len(y)))
        same = (ref.view(np.uint32) == got.view(np.uint32)) | (np.isnan(ref) & np.isnan(got))
        assert same.all(), (y[~same][:5], x[~same][:5], ref[~same][:5], got[~same][:5])
