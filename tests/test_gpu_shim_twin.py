"""The bodies of the Rust shim, run.

akaze-rust_amd/rust/src/lib.rs cannot be compiled in this image.  tests/shim_twin/shim_twin.cpp restates every shim
function body in C++ (same C-ABI calls, same order, same host arithmetic); tools/check_shim.py (a CPU test) holds the two
to the same call lists.  Here the twin runs on the GPU and every function is compared with the oracle, as a test of the
reference crate's own functions would read."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TWIN_PATH = os.path.join(HERE, "shim_twin", "libshim_twin.so")
PLANES = ["Lt", "Lsmooth", "Lx", "Ly", "Lxx", "Lyy", "Lxy", "Lflow", "Lstep", "Ldet"]


@pytest.fixture(scope="module")
def twin(amd):
    assert os.path.exists(TWIN_PATH), "tests/shim_twin/libshim_twin.so is built by __graft_entry__.build()"
    L = C.CDLL(TWIN_PATH)
    L.twin_last_error.restype = C.c_char_p
    vp, u64, u32, i32, f32, f64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_float, C.c_double
    # (arguments past the sixth travel on the stack, where an undeclared Python int fills only half of a 64-bit slot)
    for name, args in {
        "twin_allocate_evolutions": [u32, u32, vp, u64, vp, vp, vp, vp, vp, u64],
        "twin_contrast_factor": [vp, u32, u32, f64, f64, u64, vp],
        "twin_fed_tau": [f64, i32, f64, i32, vp, u64, vp],
        "twin_descriptor_match": [vp, u64, vp, u64, u64, u64, f64, vp, vp],
        "twin_remove_outliers": [vp, u64, vp, u64, vp, u64, u64, f32, f32, vp, vp],
        "twin_estimate_fundamental_matrix": [vp, u64, vp, u64, vp, f32, vp, vp],
        "twin_match_features": [vp, u64, vp, u64, vp, u64, vp, u64, u64, f64, u64, f32, vp, vp],
        "twin_draw_keypoints": [vp, u32, u32, vp, u64, vp],
        "twin_draw_matches": [vp, u32, u32, vp, u32, u32, vp, u64, vp, u64, vp, u64, vp, u64, vp, vp],
        "twin_features_level": [vp, u64, vp, vp, vp, vp, vp],
        "twin_features_plane": [vp, u64, i32, vp, vp],
        "twin_detect_keypoints": [vp, vp, vp, u64, vp],
        "twin_extract_descriptors": [vp, vp, vp, u64, vp],
        "twin_features_free": [vp],
    }.items():
        getattr(L, name).argtypes = args
    return L


def ok(L, status):
    assert status == 0, L.twin_last_error().decode()


def fp(a):
    return a.ctypes.data_as(C.c_void_p)


def ref_of(obj):
    """a ctypes object's address as c_void_p (what declared c_void_p parameters accept)"""
    return C.cast(C.pointer(obj), C.c_void_p)


def rand_img(h, w, seed, lo=0.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, (h, w)).astype(np.float32)


def same(a, b):
    assert a.shape == b.shape and np.array_equal(a, b), (a.shape, b.shape, np.argwhere(a != b)[:3] if a.shape == b.shape else None)


@pytest.mark.parametrize("shape", [(37, 53), (64, 65), (101, 259)])
def test_image_functions(twin, ref, shape):
    h, w = shape
    img = rand_img(h, w, 1, -1.0)
    out = np.empty((h // 2, w // 2), np.float32)
    ok(twin, twin.twin_half_size(fp(img), w, h, fp(out)))
    same(out, ref.half_size(img))
    for klen in (3, 5, 9):
        k = rand_img(1, klen, 2)[0].copy()
        for horizontal, fn in ((1, ref.horizontal_filter), (0, ref.vertical_filter)):
            o = np.empty_like(img)
            ok(twin, twin.twin_filter(fp(img), w, h, fp(k), klen, horizontal, fp(o)))
            same(o, fn(img, k))
    o = np.empty_like(img)
    ok(twin, twin.twin_gaussian_blur(fp(img), w, h, C.c_float(1.6), fp(o)))
    same(o, ref.gaussian_blur(img, 1.6))
    for xo, yo, s in ((1, 0, 1), (0, 1, 2), (1, 1, 3), (0, 0, 1), (1, 1, 1)):
        ok(twin, twin.twin_scharr(fp(img), w, h, xo, yo, s, fp(o)))
        same(o, ref.scharr(img, bool(xo), bool(yo), s))
    k = C.c_double()
    pos = rand_img(h, w, 3)
    ok(twin, twin.twin_contrast_factor(fp(pos), w, h, C.c_double(0.7), C.c_double(1.0), 300, ref_of(k)))
    assert k.value == ref.contrast_factor(pos, 0.7, 1.0, 300)


def test_host_side_image_helpers(twin):
    """fill_border, sqrt_squared, normalize, the two 8-bit conversions: host code of the shim (image.rs:127-260)"""
    h, w = 23, 31
    img = rand_img(h, w, 4, -2.0, 3.0)
    for hw in (1, 3, 11, 12):
        got = img.copy()
        ok(twin, twin.twin_fill_border(fp(got), w, h, hw))
        if w <= 2 * hw or h <= 2 * hw:
            same(got, img)
            continue
        ys = np.clip(np.arange(h), hw, h - 1 - hw)
        xs = np.clip(np.arange(w), hw, w - 1 - hw)
        same(got, img[np.ix_(ys, xs)])  # image.rs:239-260: border pixels take the nearest interior value
    a, b = rand_img(h, w, 5), rand_img(h, w, 6)
    got = a.copy()
    ok(twin, twin.twin_sqrt_squared(fp(got), fp(b), w, h))
    same(got, a + b)  # image.rs:218-231 (sic)
    got = np.empty_like(img)
    ok(twin, twin.twin_normalize(fp(img), w, h, fp(got)))
    same(got, (img - img.min()) / (img.max() - img.min()))
    luma = np.random.default_rng(7).integers(0, 256, (h, w), dtype=np.uint8)
    f = np.empty((h, w), np.float32)
    ok(twin, twin.twin_unit_float(fp(luma), w, h, fp(f)))
    same(f, luma.astype(np.float32) * np.float32(1) / np.float32(255))
    edge = np.array([[-1.0, 0.0, 0.5, 1.0, 1.5, np.nan, 0.999]], np.float32)
    u = np.empty(edge.shape, np.uint8)
    ok(twin, twin.twin_dynamic_image(fp(edge), edge.shape[1], 1, fp(u)))
    assert u.tolist() == [[0, 0, 127, 255, 255, 0, 254]]  # `(v * 255f32) as u8`: saturating, NaN -> 0


def test_allocate_evolutions_and_fed_tau(twin, amd, ref):
    for w, h, kw in ((1920, 1080, {}), (159, 79, {}), (3840, 2160, dict(num_sublevels=5, max_octave_evolution=5))):
        cfg = amd.Config(**kw)
        n = C.c_uint64()
        times = np.zeros((64, 2)); ints = np.zeros((64, 3), np.uint32); n_tau = np.zeros(64, np.uint64); tau = np.zeros(4096)
        ok(twin, twin.twin_allocate_evolutions(w, h, ref_of(cfg), 64, ref_of(n), fp(times), fp(ints), fp(n_tau), fp(tau), 4096))
        plan = amd.plan_levels(w, h, cfg)
        assert n.value == len(plan)
        at = 0
        for i, lv in enumerate(plan):
            assert (times[i, 0], times[i, 1]) == (lv["etime"], lv["esigma"])
            assert tuple(ints[i]) == (lv["octave"], lv["sublevel"], lv["sigma_size"])
            assert tau[at:at + int(n_tau[i])].tobytes() == np.asarray(lv["tau"], np.float64).tobytes()
            at += int(n_tau[i])
    out = np.zeros(64); n = C.c_uint64()
    ok(twin, twin.twin_fed_tau(C.c_double(5.9984531212995087), 1, C.c_double(0.25), 1, fp(out), 64, ref_of(n)))
    want = ref.fed_tau(5.9984531212995087)
    assert n.value == len(want) == 8 and out[:8].tobytes() == np.asarray(want).tobytes()
    # where the reference never terminates the shim panics: status -1 with the library's message
    assert twin.twin_fed_tau(C.c_double(0.1), 1, C.c_double(0.25), 1, fp(out), 64, ref_of(n)) == -1
    assert b"akaze_hip status" in twin.twin_last_error()


def test_calculate_step_and_eval(twin, ref):
    h, w = 45, 67
    lt, lflow = rand_img(h, w, 8), rand_img(h, w, 9)
    got, lstep = lt.copy(), np.empty_like(lt)
    ok(twin, twin.twin_calculate_step(fp(got), fp(lflow), w, h, C.c_double(0.35), fp(lstep)))
    want, want_step = ref.fed_step(lt, lflow, 0.35)
    same(got, want)
    same(lstep, want_step)
    # nonlinear_diffusion.rs:149-173 with the offsets of the interior case (:63-67, xpos)
    px = np.array([0, 1, 1, 0], np.int32); py = np.zeros(4, np.int32)
    v = C.c_float()
    ok(twin, twin.twin_eval(fp(lflow), fp(lt), w, h, 10, 20, fp(px), fp(py), ref_of(v)))
    assert np.float32(v.value) == (lflow[20, 10] + lflow[20, 11]) * (lt[20, 11] - lt[20, 10])


@pytest.fixture(scope="module")
def features(twin, amd):
    frame = amd.synth_frame(640, 480, 1)
    cfg = amd.Config()
    h = C.c_void_p()
    ok(twin, twin.twin_extract_features(fp(frame), 640, 480, ref_of(cfg), ref_of(h)))
    yield frame, cfg, h
    twin.twin_features_free(h)


def _kp(twin, amd, h):
    nl, nk, nb = C.c_uint64(), C.c_uint64(), C.c_uint64()
    twin.twin_features_counts(h, ref_of(nl), ref_of(nk), ref_of(nb))
    kp = np.zeros(nk.value, amd.KEYPOINT_DTYPE)
    ok(twin, twin.twin_features_keypoints(h, fp(kp)))
    d = np.zeros((nk.value, nb.value), np.uint8)
    ok(twin, twin.twin_features_descriptors(h, fp(d)))
    return nl.value, kp, d


def test_extract_features_returns_what_the_reference_returns(twin, amd, ref, features):
    frame, cfg, h = features
    rf = ref.extract(frame)
    nl, kp, d = _kp(twin, amd, h)
    assert nl == rf.num_levels and len(kp) == rf.num_keypoints > 50
    rk = rf.keypoints()
    for f in ("x", "y", "response", "size", "octave", "class_id", "angle"):
        assert np.array_equal(kp[f], rk[f]), f
    assert np.array_equal(d, rf.descriptors())
    for lvl in range(nl):
        times = np.zeros(2); ints = np.zeros(3, np.uint32); wh = np.zeros(2, np.uint32); nt = C.c_uint64(); tau = np.zeros(8192)
        twin.twin_features_level(h, lvl, fp(times), fp(ints), fp(wh), ref_of(nt), fp(tau))
        info = rf.level_info(lvl)
        assert (times[0], times[1], *ints, *wh) == (info["etime"], info["esigma"], info["octave"], info["sublevel"],
                                                     info["sigma_size"], info["w"], info["h"])
        assert tau[:nt.value].tobytes() == info["tau"].tobytes()
        for pi, name in enumerate(PLANES):
            n_px = C.c_uint64()
            twin.twin_features_plane(h, lvl, pi, None, ref_of(n_px))
            want = rf.plane(lvl, name)
            assert n_px.value == want.size, (lvl, name)  # level 0: Lflow, Lstep are 0 x 0
            if want.size:
                got = np.empty(want.shape, np.float32)
                twin.twin_features_plane(h, lvl, pi, fp(got), ref_of(n_px))
                same(got, want)


def test_ops_on_the_callers_evolutions(twin, amd, ref, features):
    """detect_keypoints / extract_descriptors / detector_response take `&[EvolutionStep]` the caller holds: the shim
    sends those planes back through akz_extract_from_planes (scale_space_extrema.rs:199-203, descriptors.rs:14-27,
    detector_response.rs:38-55)"""
    frame, cfg, h = features
    rf = ref.extract(frame)
    _, kp, d = _kp(twin, amd, h)
    ok(twin, twin.twin_detector_response(h, ref_of(cfg)))  # recomputed from Lsmooth, must reproduce the planes
    for lvl in (0, 5, rf.num_levels - 1):
        for pi in (2, 3, 4, 5, 6, 9):
            want = rf.plane(lvl, PLANES[pi])
            got = np.empty(want.shape, np.float32); n_px = C.c_uint64()
            twin.twin_features_plane(h, lvl, pi, fp(got), ref_of(n_px))
            same(got, want)
    out = np.zeros(len(kp) + 16, amd.KEYPOINT_DTYPE); n = C.c_uint64()
    ok(twin, twin.twin_detect_keypoints(h, ref_of(cfg), fp(out), len(out), ref_of(n)))
    assert n.value == len(kp) and out[:len(kp)].tobytes() == kp.tobytes()
    sub = np.ascontiguousarray(kp[::3])
    dd = np.zeros((len(sub), 61), np.uint8)
    ok(twin, twin.twin_extract_descriptors(h, ref_of(cfg), fp(sub), len(sub), fp(dd)))
    assert np.array_equal(dd, d[::3])
    assert twin.twin_extract_descriptors(h, ref_of(cfg), fp(sub), 0, fp(dd)) == 0  # descriptors.rs: no keypoints, no work


def test_matching_functions(twin, amd, ref):
    f0, f1 = amd.synth_frame(480, 360, 0), amd.synth_frame(480, 360, 0, shift=(7, 4))
    r0, r1 = ref.extract(f0), ref.extract(f1)
    d0, d1 = r0.descriptors(), r1.descriptors()
    k0, k1 = r0.keypoints().astype(amd.KEYPOINT_DTYPE), r1.keypoints().astype(amd.KEYPOINT_DTYPE)
    want = ref.descriptor_match(d0, d1, 10000, 0.86)
    out = np.zeros(len(d0), amd.MATCH_DTYPE); n = C.c_uint64()
    ok(twin, twin.twin_descriptor_match(fp(d0), len(d0), fp(d1), len(d1), 61, 10000, C.c_double(0.86), fp(out), ref_of(n)))
    assert n.value == len(want) > 50 and out[:n.value].tobytes() == np.ascontiguousarray(want).tobytes()
    # empty sides (the reference returns an empty Vec)
    ok(twin, twin.twin_descriptor_match(fp(d0), 0, fp(d1), len(d1), 61, 10000, C.c_double(0.86), fp(out), ref_of(n)))
    assert n.value == 0
    # RANSAC from the pinned random stream: the same stream in the oracle
    ref.random_seed(42, 69); amd.random_seed(42, 69)
    want_in = ref.remove_outliers(r0.keypoints(), r1.keypoints(), want, 200, 0.05, 3.0)
    got = np.zeros(len(want), amd.MATCH_DTYPE)
    ok(twin, twin.twin_remove_outliers(fp(k0), len(k0), fp(k1), len(k1), fp(np.ascontiguousarray(want)), len(want), 200,
                                       C.c_float(0.05), C.c_float(3.0), fp(got), ref_of(n)))
    assert n.value == len(want_in) and got[:n.value].tobytes() == np.ascontiguousarray(want_in).tobytes()
    ref.random_seed(42, 69); amd.random_seed(42, 69)
    want_in = ref.remove_outliers(r0.keypoints(), r1.keypoints(), want, 200, 0.05, 3.0)
    ok(twin, twin.twin_match_features(fp(k0), len(k0), fp(d0), len(d0), fp(k1), len(k1), fp(d1), len(d1), 61, C.c_double(0.86), 200,
                                      C.c_float(3.0), fp(got), ref_of(n)))
    assert n.value == len(want_in) and got[:n.value].tobytes() == np.ascontiguousarray(want_in).tobytes()
    f9 = np.zeros(9, np.float32); found = C.c_int()
    m8 = np.ascontiguousarray(want[:8])
    ok(twin, twin.twin_estimate_fundamental_matrix(fp(k0), len(k0), fp(k1), len(k1), fp(m8), C.c_float(0.05), fp(f9), ref_of(found)))
    F = amd.estimate_fundamental_matrix(k0, k1, m8, 0.05)
    assert (found.value != 0) == (F is not None) and (F is None or np.array_equal(np.asarray(F, np.float32).ravel(), f9))


def test_drawing_functions(twin, amd):
    """the shim's drawing functions against the product's own entry points (which tests/test_reference_outputs.py and
    test_gpu_cli.py hold to the reference's published pictures)"""
    rng = np.random.default_rng(11)
    rgb = rng.integers(0, 256, (90, 120, 3), dtype=np.uint8)
    kp = np.zeros(12, amd.KEYPOINT_DTYPE)
    kp["x"], kp["y"], kp["size"] = rng.uniform(5, 115, 12), rng.uniform(5, 85, 12), rng.uniform(2, 9, 12)
    amd.random_seed(42, 69)
    want = amd.draw_keypoints(rgb, kp)
    amd.random_seed(42, 69)
    got = np.empty_like(rgb)
    ok(twin, twin.twin_draw_keypoints(fp(rgb), 120, 90, fp(kp), len(kp), fp(got)))
    assert np.array_equal(got, want) and not np.array_equal(got, rgb)
    m = np.zeros(5, amd.MATCH_DTYPE); m["index_0"] = np.arange(5); m["index_1"] = np.arange(5)[::-1]
    amd.random_seed(42, 69)
    want = amd.draw_matches(rgb, rgb[:, :100].copy(), kp, kp, m)
    amd.random_seed(42, 69)
    got = np.empty(want.size, np.uint8); ow, oh = C.c_uint32(), C.c_uint32()
    second = np.ascontiguousarray(rgb[:, :100])
    ok(twin, twin.twin_draw_matches(fp(rgb), 120, 90, fp(second), 100, 90, fp(kp), len(kp), fp(kp), len(kp), fp(m), len(m), fp(got),
                                    got.size, ref_of(ow), ref_of(oh)))
    assert (oh.value, ow.value) == want.shape[:2] and np.array_equal(got.reshape(want.shape), want)
    c = np.zeros(3, np.uint8)
    amd.random_seed(42, 69)
    ok(twin, twin.twin_random_color(fp(c)))
    img, want = np.zeros((40, 40, 3), np.uint8), np.zeros((40, 40, 3), np.uint8)
    L = amd.lib()
    ok(twin, twin.twin_draw_circle(fp(img), 40, 40, C.c_float(20), C.c_float(20), fp(c), C.c_float(5)))
    assert L.akz_draw_circle(fp(want), 40, 40, C.c_float(20), C.c_float(20), fp(c), C.c_float(5)) == 0
    assert img.any() and np.array_equal(img, want)
    ok(twin, twin.twin_draw_line(fp(img), 40, 40, C.c_float(2), C.c_float(2), C.c_float(30), C.c_float(35), fp(c), C.c_float(1)))
    assert L.akz_draw_line(fp(want), 40, 40, C.c_float(2), C.c_float(2), C.c_float(30), C.c_float(35), fp(c), C.c_float(1)) == 0
    assert np.array_equal(img, want)
