"""The DEVICE build of csrc/akz_libm.hpp (atan2f / cosf / sinf as this machine's glibc computes them) against the host's libm,
and what it buys: a job whose keypoint selection ran on the device forms its angles and descriptors without a host round
trip -- with the same keypoints, angles and descriptor bytes as with the host's libm, and as the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hostlibm():
    path = os.path.join(HERE, "libm_check", "liblibm_check.so")
    assert os.path.exists(path), "tests/libm_check/liblibm_check.so is built by __graft_entry__.build()"
    return C.CDLL(path)


def fp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_device_forms_equal_the_hosts_libm(ctx, hostlibm):
    import torch
    avail, _ = ctx.debug_device_libm()
    assert avail in (1, 2), "on this image (x86-64, glibc 2.35) the self-test must pass"
    rng = np.random.default_rng(9)
    # sinf / cosf: every third float of [2^-13, 120), both signs, + everything tiny (stride)
    pos = np.arange(0x39000000, 0x42f00000, 3, dtype=np.uint32)
    small = np.arange(0, 0x39000000, 1021, dtype=np.uint32)
    a = np.concatenate([pos, pos | 0x80000000, small, small | 0x80000000]).view(np.float32)
    b = (2.0 ** rng.uniform(-60, 60, len(a))).astype(np.float32) * rng.choice(np.float32([-1, 1]), len(a))
    sw = rng.random(len(a)) < 0.5   # atan2f sees (a, b) and (b, a)
    y, x = np.where(sw, a, b), np.where(sw, b, a)
    step = 1 << 25
    for lo in range(0, len(a), step):
        ya, xa, aa = (np.ascontiguousarray(v[lo:lo + step]) for v in (y, x, a))
        got_at = ctx.debug_libm_eval(torch.from_numpy(ya).cuda(), torch.from_numpy(xa).cuda(), fma=avail == 1)[:, 0].cpu().numpy()
        got_cs = ctx.debug_libm_eval(torch.from_numpy(aa).cuda(), torch.from_numpy(xa).cuda(), fma=avail == 1)[:, 1:].cpu().numpy()
        ref = np.empty_like(ya)
        hostlibm.lc_atan2f(0, fp(ya), fp(xa), fp(ref), C.c_uint64(len(ya)))
        assert (ref.view(np.uint32) == got_at.view(np.uint32)).all()
        for col, fn in ((0, hostlibm.lc_cosf), (1, hostlibm.lc_sinf)):
            fn(0, fp(aa), fp(ref), C.c_uint64(len(aa)), None)
            g = np.ascontiguousarray(got_cs[:, col])
            bad = np.nonzero(ref.view(np.uint32) != g.view(np.uint32))[0]
            assert len(bad) == 0, (aa[bad[:4]], ref[bad[:4]], g[bad[:4]])


@pytest.mark.parametrize("w,h,n", [(1920, 1080, 1), (640, 480, 3), (3840, 2160, 1)])
def test_angles_and_descriptors_without_the_host_round_trip(ctx, amd, ref, w, h, n):
    """scale_space_extrema.rs:326 (atan2), descriptors.rs:55-56 (cos, sin): on the device behind the device's selection -- same
    result as with the host's libm, and as the oracle's"""
    import torch
    frames = np.stack([amd.synth_frame(w, h, 70 + i) for i in range(n)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    try:
        ctx.debug_set_select(2)            # the selection on the device (what a waited-for job takes by itself)
        ctx.debug_set_device_libm(True)
        dev = ctx.extract_features(d, keep_all_planes=False)
        assert ctx.debug_device_libm()[1] in (1, 2)     # this job's angles came from the device
        dev2 = ctx.extract_begin(d, amd.Config(), keep_all_planes=False, host_descriptors=False).finish()   # descriptors stay on the device
        ctx.debug_set_device_libm(False)
        host = ctx.extract_features(d, keep_all_planes=False)
        assert ctx.debug_device_libm()[1] == 0
        for img in range(n):
            kd, kh = dev.keypoints(img), host.keypoints(img)
            assert len(kd) == len(kh) > 50 and kd.tobytes() == kh.tobytes()
            assert np.array_equal(dev.descriptors(img), host.descriptors(img))
            assert dev2.keypoints(img).tobytes() == kh.tobytes()
            total = sum(dev2.counts(i)[1] for i in range(n))
            allrows = torch.empty((total, 64), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()   # (the copy runs on a stream of the context's own: nothing of torch's may still be writing the buffer)
            dev2.copy_device_descriptors(allrows)
            first = sum(dev2.counts(i)[1] for i in range(img))
            rows = allrows[first:first + len(kh)].cpu().numpy()
            assert np.array_equal(rows[:, :61], host.descriptors(img)) and not rows[:, 61:].any()
            if w <= 1920:
                rf = ref.extract(frames[img], threads=8)
                assert np.array_equal(kd["angle"], rf.keypoints()["angle"]) and np.array_equal(dev.descriptors(img), rf.descriptors())
                rf.close()
        for r in (dev, dev2, host):
            r.close()
    finally:
        ctx.debug_set_select(None)
        ctx.debug_set_device_libm(True)
