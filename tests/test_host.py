"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol that
include/akaze_hip.h declares, host planning agrees with the oracle, and the HIP path fails loudly
without a GPU.  No compute calls."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    syms = set()
    for name in ("akaze_hip.h", "akaze_hip_debug.h"):  # the drop-in boundary and the measurement / test hooks
        hdr = open(os.path.join(ROOT, "include", name)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        syms |= set(re.findall(r"\b(akz_[a-z0-9_]+)\s*\(", hdr))
    return sorted(syms)


def test_debug_hooks_are_not_in_the_public_header():
    hdr = open(os.path.join(ROOT, "include", "akaze_hip.h")).read()
    for s in ("akz_ctx_graph_probe", "akz_debug_", "akz_fed_kernel_name", "akz_detector_kernel_name", "akz_ctx_set_fed_mode",
              "akz_ctx_set_match_mode", "akz_ctx_set_detector_mode", "akz_ctx_set_prep_mode", "akz_synth_frame_u8"):
        assert s not in hdr, s


def test_library_exports_every_declared_symbol(amd):
    L = amd.lib()
    syms = declared_symbols()
    assert len(syms) >= 45
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing
    # and the binding declares a signature for each of them
    assert sorted(set(syms) - set(L._declared)) == []
    hdr = open(os.path.join(ROOT, "include", "akaze_hip.h")).read()
    assert L.akz_abi_version() == int(re.search(r"#define\s+AKZ_ABI_VERSION\s+(\d+)", hdr).group(1)) == 6


def test_no_oracle_in_product_path():
    """The product must not reference the oracle (a CPU fallback would void parity claims)."""
    pkg = os.path.join(ROOT, "akaze-rust_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".cpp", ".hpp", ".hip", ".py", ".h", ".rs")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "akaze_ref" not in txt and "oracle/" not in txt, os.path.join(dirpath, f)
    out = subprocess.check_output(["ldd", os.path.join(pkg, "libakaze_hip.so")]).decode()
    assert "akaze_ref" not in out


def test_config_default_matches_reference(amd, ref):
    c, r = amd.Config(), ref.default_config()
    for name, _ in amd.Config._fields_:
        assert getattr(c, name) == getattr(r, name), name
    # akaze/src/types/evolution.rs:40-55
    assert (c.num_sublevels, c.max_octave_evolution, c.base_scale_offset) == (4, 4, 1.6)
    assert (c.contrast_percentile, c.contrast_factor_num_bins, c.derivative_factor) == (0.7, 300, 1.5)
    assert (c.detector_threshold, c.descriptor_channels, c.descriptor_pattern_size) == (0.001, 3, 10)


def test_struct_layouts(amd):
    assert C.sizeof(amd.Config) == 72
    assert amd.KEYPOINT_DTYPE.itemsize == 40 and amd.MATCH_DTYPE.itemsize == 24


@pytest.mark.parametrize("T", [0.53019335983756166, 1.0603867196751229, 5.9984531212995087, 23.7, 400.0])
def test_fed_tau_matches_oracle(amd, ref, T):
    a, b = amd.fed_tau_by_process_time(T), ref.fed_tau(T)
    assert a.tobytes() == b.tobytes()
    a, b = amd.fed_tau_by_process_time(T, 2, 0.25, False), ref.fed_tau(T, 2, 0.25, False)
    assert a.tobytes() == b.tobytes()


def test_fed_tau_nonterminating_case_is_an_error(amd):
    with pytest.raises(amd.AkazeError) as e:
        amd.fed_tau_by_process_time(0.1)  # n == 1 (fed_tau.rs:95)
    assert e.value.status == -6


def test_kernels_match_oracle(amd, ref):
    for sigma, size in [(3.0, 7), (1.6, 5), (1.0, 3), (2.5, 7)]:
        assert amd.gaussian_kernel(sigma, size).tobytes() == ref.gaussian_kernel(sigma, size).tobytes()
    for s in (1, 2, 3, 4, 5):
        for a, b in zip(amd.scharr_kernels(s), ref.scharr_kernels(s)):
            assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("w,h,kw", [(1920, 1080, {}), (3840, 2160, {}), (2016, 1512, {}), (320, 240, {}),
                                    (3840, 2160, dict(num_sublevels=5, max_octave_evolution=5)), (100, 50, {})])
def test_plan_matches_oracle(amd, ref, w, h, kw):
    plan = amd.plan_levels(w, h, amd.Config(**kw))
    # the oracle's planner runs inside extract(); drive it with a flat frame of the smallest size
    # that yields the same octave/sublevel table, and compare the size-independent fields.
    r = ref.extract(np.full((h // 8, w // 8), 9, np.uint8), ref.default_config(**kw)) if min(w, h) >= 640 else \
        ref.extract(np.full((h, w), 9, np.uint8), ref.default_config(**kw))
    scale = 8 if min(w, h) >= 640 else 1
    oct_lim = r.num_levels
    for lvl, p in enumerate(plan[:oct_lim]):
        q = r.level_info(lvl)
        for f in ("etime", "esigma", "octave", "sublevel", "sigma_size"):
            assert p[f] == q[f], (lvl, f)
        assert p["tau"].tobytes() == q["tau"].tobytes()
        if scale == 1:
            assert (p["w"], p["h"]) == (q["w"], q["h"])
    if (w, h) == (1920, 1080):
        assert [len(p["tau"]) for p in plan] == [0, 3, 3, 4, 4, 5, 6, 7, 8, 10, 12, 14, 17, 20, 24, 29]
        assert [p["det_sigma"] for p in plan] == [2, 3, 3, 4] * 4
        assert [(p["w"], p["h"]) for p in plan[::4]] == [(1920, 1080), (960, 540), (480, 270), (240, 135)]
    if kw:
        assert len(plan) == 25 and sum(len(p["tau"]) for p in plan) == 400


def test_synth_frame_is_deterministic(amd):
    a, b = amd.synth_frame(200, 120, 3), amd.synth_frame(200, 120, 3)
    assert np.array_equal(a, b) and a.std() > 10
    assert not np.array_equal(a, amd.synth_frame(200, 120, 4))


def test_fails_loudly_without_gpu(amd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    with pytest.raises(amd.AkazeError) as e:
        amd.Context(0)
    assert e.value.status == -3 and "no CPU fallback" in str(e.value)


def test_design_gate_table_is_the_librarys(amd):
    """DESIGN.md 6.1 is rendered from the library's own gate table (csrc/akz_gates.hpp through akz_debug_gates)"""
    p = subprocess.run([__import__("sys").executable, os.path.join(ROOT, "tools", "render_gates.py"), "--check"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    names = [g["name"] for g in amd.gates()]
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert all(f"`{n}`" in design for n in names) and len(names) >= 12


def test_isa_has_no_contracted_fma():
    """Bit-exact parity needs un-fused mul/add in the image arithmetic (tools/isa_audit.py)."""
    pkg = os.path.join(ROOT, "akaze-rust_amd")
    subprocess.check_call(["make", "-C", pkg, "asm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_audit
    rows, bad = isa_audit.audit(os.path.join(pkg, "csrc", "akz_kernels.s"))
    rows2, bad2 = isa_audit.audit(os.path.join(pkg, "csrc", "akz_stencil.s"))
    assert len(rows) >= 14 and len(rows2) >= 18 and not bad and not bad2, (bad, bad2)
    # k_octave_resident: f64 FMAs only, and only those of pm_g2 (akz_pm_g2.hpp): per pixel site the four of the refined
    # reciprocal (one v_rcp_f64) plus the five of the full division's expansion on the rare path (a second v_rcp_f64, one
    # v_div_fmas_f64); a plain division (1 / k^2) has the expansion alone.  No f32 FMA at all (the 2x2 mean's division by 4
    # is a multiplication by 0.25)
    import re
    res = open(os.path.join(pkg, "csrc", "akz_resident.s")).read()
    assert not re.findall(r"\bv_(?:pk_fma|fma|fmac|mad|mac|fmaak|fmamk)_(?:f32|legacy_f32)\b", res)
    n_div = len(re.findall(r"\bv_div_fmas_f64\b", res))
    n_rcp = len(re.findall(r"\bv_rcp_f64", res))
    sites, plain = n_rcp - n_div, 2 * n_div - n_rcp
    assert sites >= 2 and plain >= 0 and len(re.findall(r"\bv_(?:fma|fmac)_f64", res)) == 9 * sites + 5 * plain
    pure = {r[0]: r[1] for r in rows + rows2}
    for k, v in pure.items():
        if k.startswith("k_fed_own") and "Lb1E" in k:
            continue  # the variants that prepare the next level as their epilogue carry pm_g2: checked below
        if k.startswith(("k_fed", "k_filter_v", "k_filter_hIf", "k_ldet", "k_nms", "k_orientation", "k_deriv",
                         "k_blurILi1EfE", "k_blurILi2EfE")):
            assert v == 0, (k, v)
    # k_prep and k_fed_own's epilogue (the same akz_prep_passes.hpp): the same account -- f64 FMAs of pm_g2's one pixel site and
    # of 1 / k^2, no f32 FMA
    ker = open(os.path.join(pkg, "csrc", "akz_kernels.s")).read() + open(os.path.join(pkg, "csrc", "akz_stencil.s")).read()
    seen = 0
    for m in re.finditer(r"^(_ZN3akz\w+):[^\n]*\n(.*?)^\.Lfunc_end", ker, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not (("k_fed_own" in name and "Lb1E" in name) or "k_prepILb" in name):
            continue
        seen += 1
        assert not re.findall(r"\bv_(?:pk_fma|fma|fmac|mad|mac|fmaak|fmamk)_(?:f32|legacy_f32)\b", body), name
        n_div = len(re.findall(r"\bv_div_fmas_f64\b", body))
        n_rcp = len(re.findall(r"\bv_rcp_f64", body))
        # (the passes are compiled twice -- tiles in the image's interior without tests and clamps, the others with: two pixel sites)
        sites, plain = n_rcp - n_div, 2 * n_div - n_rcp
        assert sites in (1, 2) and plain == 1 and len(re.findall(r"\bv_(?:fma|fmac)_f64", body)) == 9 * sites + 5 * plain, name
    assert seen == 4, seen


def _round_half_away(v):
    v = np.asarray(v, np.float32)
    return np.trunc(v + np.copysign(np.float32(0.5), v)).astype(np.float32)


def nms_candidates_numpy(r, cfg):
    """numpy statement of what k_nms hands to the host: threshold + strict 4-neighbour maximum
    (scale_space_extrema.rs:32-42) + the border test (:80-87), any order."""
    import akaze_amd
    out = []
    thr = np.float32(cfg.detector_threshold)
    smax = np.float32(10.0) * np.sqrt(np.float32(2.0))
    for lvl in range(r.num_levels):
        info = r.level_info(lvl)
        D = r.plane(lvl, "Ldet")
        h, w = D.shape
        size = np.float32(info["esigma"] * cfg.derivative_factor)
        ratio = np.float32(2.0 ** info["octave"])
        m = smax * _round_half_away(size / ratio)
        c = D[1:-1, 1:-1]
        ok = (c > thr) & (c > D[1:-1, 2:]) & (c > D[1:-1, :-2]) & (c > D[:-2, 1:-1]) & (c > D[2:, 1:-1])
        ys, xs = np.nonzero(ok)
        ys, xs = ys + 1, xs + 1
        fx, fy = xs.astype(np.float32), ys.astype(np.float32)
        is_out = ((_round_half_away(fx - m) - 1 < 0) | (_round_half_away(fx + m) + 1 >= np.float32(w)) |
                  (_round_half_away(fy - m) - 1 < 0) | (_round_half_away(fy + m) + 1 >= np.float32(h)))
        for x, y in zip(xs[~is_out], ys[~is_out]):
            out.append((lvl, y * w + x, D[y, x], D[y, x + 1], D[y, x - 1], D[y + 1, x], D[y - 1, x], 0))
    return np.array(out, akaze_amd.CANDIDATE_DTYPE)


@pytest.mark.parametrize("w,h,idx,kw", [(640, 480, 1, {}), (960, 540, 3, {}),
                                        (800, 600, 2, dict(detector_threshold=0.0002))])
def test_host_keypoint_logic_matches_oracle(amd, ref, w, h, idx, kw):
    """The order-dependent cache logic + sub-pixel step run on the host (akz_keypoints.cpp) must give the
    oracle's keypoints when fed the candidates the NMS kernel would emit (here computed with numpy from
    the oracle's Ldet planes, in shuffled order)."""
    frame = amd.synth_frame(w, h, idx)
    r = ref.extract(frame, ref.default_config(**kw))
    cfg = amd.Config(**kw)
    cands = nms_candidates_numpy(r, cfg)
    assert len(cands) > 50
    rng = np.random.default_rng(0)
    kps, n_ext = amd.host_select_keypoints(w, h, cfg, cands[rng.permutation(len(cands))])
    rk = r.keypoints()
    assert n_ext == r.num_extrema and len(kps) == len(rk)
    for f in ("x", "y", "response", "size", "octave", "class_id"):
        assert np.array_equal(kps[f], rk[f]), f


def _two_view_scene(n, n_outliers, seed):
    """n points seen by two cameras (a translation + small rotation): exact epipolar geometry, plus
    n_outliers wrong correspondences at the END of the match list."""
    import akaze_amd
    rng = np.random.default_rng(seed)
    X = np.c_[rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(6, 14, n)]
    f, cx, cy = 900.0, 960.0, 540.0
    th = 0.05
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    t = np.array([0.6, 0.05, 0.1])
    def proj(P):
        return np.c_[f * P[:, 0] / P[:, 2] + cx, f * P[:, 1] / P[:, 2] + cy]
    p0, p1 = proj(X), proj(X @ R.T + t)
    k0 = np.zeros(n, akaze_amd.KEYPOINT_DTYPE); k1 = np.zeros(n, akaze_amd.KEYPOINT_DTYPE)
    k0["x"], k0["y"] = p0[:, 0], p0[:, 1]
    k1["x"], k1["y"] = p1[:, 0], p1[:, 1]
    m = np.zeros(n, akaze_amd.MATCH_DTYPE)
    m["index_0"] = np.arange(n); m["index_1"] = np.arange(n); m["distance"] = rng.integers(0, 80, n)
    if n_outliers:
        wrong = np.arange(n - n_outliers, n)
        m["index_1"][wrong] = (wrong * 7 + 3) % (n - n_outliers)
    return k0, k1, m


def test_remove_outliers_concurrent_callers(amd):
    """The trials of a call run on a process-wide pool of host threads; callers that arrive while it is busy start their
    own threads.  Every caller thread has its own random source: seeded alike, four concurrent callers get the result of
    a lone call."""
    import threading
    k0, k1, m = _two_view_scene(400, 60, 3)
    amd.random_seed(7, 11)
    alone = amd.remove_outliers(k0, k1, m, 400, 0.05, 0.5)
    out = [None] * 4
    def work(i):
        for _ in range(5):
            amd.random_seed(7, 11)
            out[i] = amd.remove_outliers(k0, k1, m, 400, 0.05, 0.5)
    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th: t.start()
    for t in th: t.join()
    assert all(np.array_equal(o, alone) for o in out) and len(alone) >= 8


def test_remove_outliers_host(amd, ref):
    """remove_outliers (estimate_fundamental_matrix.rs:99-165) on the host: behavioural checks and
    agreement with the oracle's restatement (different SVD method)."""
    k0, k1, m = _two_view_scene(120, 0, 1)
    # fewer than 8 matches: unchanged (:107-110)
    assert np.array_equal(amd.remove_outliers(k0, k1, m[:7], 100, 0.05, 3.0), m[:7])
    # exact geometry, no outliers.  The reference takes the right singular vector of the smallest of the
    # EIGHT singular values nalgebra returns for the 8x9 system (:46-53) — not its null vector — so even
    # perfect correspondences are not all inliers of "the model"; product and oracle must agree on that.
    # Both keep one random source per thread that persists across calls (the `random` crate's default source):
    # reseed both to the state a fresh thread starts with before every compared pair.
    def both(k0, k1, m, trials, em, ei):
        amd.random_seed(42, 69)
        ref.random_seed(42, 69)
        return amd.remove_outliers(k0, k1, m, trials, em, ei), ref.remove_outliers(k0, k1, m, trials, em, ei)
    got, exp = both(k0, k1, m, 200, 0.05, 0.5)
    assert np.array_equal(got, exp) and 8 <= len(got) <= len(m)
    # a second call continues the stream: other samples, possibly another model; reseeding repeats the first result
    again = amd.remove_outliers(k0, k1, m, 200, 0.05, 0.5)
    assert 8 <= len(again) <= len(m)
    amd.random_seed(42, 69)
    assert np.array_equal(amd.remove_outliers(k0, k1, m, 200, 0.05, 0.5), got)
    # a huge inlier tolerance keeps everything, zero trials leave the zero model, which also keeps everything
    assert np.array_equal(amd.remove_outliers(k0, k1, m, 100, 0.05, 1e9), m)
    assert np.array_equal(amd.remove_outliers(k0, k1, m, 0, 0.05, 0.5), m)
    # with wrong correspondences: output is an ordered subset that agrees with the oracle; more trials never lower
    # the best inlier count
    k0, k1, m = _two_view_scene(200, 40, 2)
    got, exp = both(k0, k1, m, 300, 0.05, 0.5)
    assert np.array_equal(got, exp)
    assert np.all(np.diff(got["index_0"].astype(np.int64)) > 0)
    assert len(got) < len(m) and set(got["index_0"].tolist()) <= set(m["index_0"].tolist())
    amd.random_seed(42, 69)
    one = amd.remove_outliers(k0, k1, m, 1, 0.05, 0.5)
    assert len(one) <= len(got)
    # degenerate sample (all points identical): rank < 8 -> no model -> zero matrix -> every match kept (:112, :152-163)
    kz = np.zeros(20, amd.KEYPOINT_DTYPE); kz["x"] = 5; kz["y"] = 7
    mz = np.zeros(20, amd.MATCH_DTYPE); mz["index_0"] = np.arange(20); mz["index_1"] = np.arange(20)
    assert np.array_equal(amd.remove_outliers(kz, kz, mz, 10, 0.05, 3.0), mz)
    # index validation instead of the reference's slice panic
    bad = m.copy(); bad["index_1"][0] = 10**6
    with pytest.raises(amd.AkazeError):
        amd.remove_outliers(k0, k1, bad, 10, 0.05, 3.0)


def test_features_and_matches_file_formats(amd, tmp_path):
    """akaze-util's on-disk formats (akaze-util/src/lib.rs:10-67): bincode 1.x bytes checked against a
    hand-built struct.pack image, serde_json schema checked with the json module, both round-trip."""
    import json
    import struct
    kps = np.zeros(3, amd.KEYPOINT_DTYPE)
    kps["x"] = [10.5, 699.0, 0.1]; kps["y"] = [20.25, 29.000069, 1e-7]; kps["response"] = [0.00101189, 0.5, 3.0]
    kps["size"] = 2.4; kps["octave"] = [0, 1, 3]; kps["class_id"] = [0, 5, 15]; kps["angle"] = [2.2353425, -0.5, 0.0]
    desc = np.random.default_rng(0).integers(0, 256, (3, 61), dtype=np.uint8)
    p = str(tmp_path / "features.cbor")  # the reference's CLIs use this suffix for bincode (lib.rs:24-28)
    amd.serialize_features_to_file(kps, desc, p)
    exp = struct.pack("<Q", 3)
    for k in kps:
        exp += struct.pack("<ffffQQf", k["x"], k["y"], k["response"], k["size"], k["octave"], k["class_id"], k["angle"])
    exp += struct.pack("<Q", 3)
    for d in desc:
        exp += struct.pack("<Q", 61) + d.tobytes()
    assert open(p, "rb").read() == exp and len(exp) == 8 + 3 * 36 + 8 + 3 * (8 + 61)
    k2, d2 = amd.deserialize_features_from_file(p)
    assert k2.tobytes() == kps.tobytes() and np.array_equal(d2, desc)
    # JSON: serde field names and nesting; floats round-trip exactly through the shortest representation
    pj = str(tmp_path / "features.json")
    amd.serialize_features_to_file(kps, desc, pj)
    doc = json.load(open(pj))
    assert list(doc) == ["keypoints", "descriptors"]
    assert list(doc["keypoints"][0]) == ["point", "response", "size", "octave", "class_id", "angle"]
    assert doc["keypoints"][1]["class_id"] == 5 and doc["descriptors"][2]["vector"] == desc[2].tolist()
    assert np.float32(doc["keypoints"][2]["point"][1]) == np.float32(1e-7)
    k3, d3 = amd.deserialize_features_from_file(pj)
    assert k3.tobytes() == kps.tobytes() and np.array_equal(d3, desc)
    # a file as serde_json would print it (different whitespace / field order must not matter)
    open(pj, "w").write('{ "descriptors": [ {"vector": [1, 2, 255]} ],\n "keypoints": [ {"angle": 0.5, "class_id": 2, '
                        '"octave": 1, "size": 4.8, "response": 0.25, "point": [3.0, 4.5]} ] }')
    k4, d4 = amd.deserialize_features_from_file(pj)
    assert (k4["x"][0], k4["y"][0], k4["class_id"][0], k4["octave"][0]) == (3.0, 4.5, 2, 1) and d4.tolist() == [[1, 2, 255]]
    # matches
    m = np.zeros(2, amd.MATCH_DTYPE)
    m["index_0"] = [1, 7]; m["index_1"] = [3, 2**40]; m["distance"] = [17.0, 0.0]
    pm = str(tmp_path / "matches.bin")
    amd.serialize_matches_to_file(m, pm)
    assert open(pm, "rb").read() == struct.pack("<Q", 2) + struct.pack("<QQd", 1, 3, 17.0) + struct.pack("<QQd", 7, 2**40, 0.0)
    assert np.array_equal(amd.deserialize_matches_from_file(pm), m)
    pmj = str(tmp_path / "matches.json")
    amd.serialize_matches_to_file(m, pmj)
    assert json.load(open(pmj)) == [{"index_0": 1, "index_1": 3, "distance": 17.0},
                                    {"index_0": 7, "index_1": 2**40, "distance": 0.0}]
    assert np.array_equal(amd.deserialize_matches_from_file(pmj), m)
    # empty sets and malformed input
    amd.serialize_features_to_file(kps[:0], desc[:0], p)
    k5, d5 = amd.deserialize_features_from_file(p)
    assert len(k5) == 0 and len(d5) == 0
    open(p, "wb").write(b"\x05\x00\x00")
    with pytest.raises(amd.AkazeError):
        amd.deserialize_features_from_file(p)


def test_estimate_fundamental_matrix_is_the_smallest_of_eight_singular_vectors(amd):
    """estimate_fundamental_matrix (estimate_fundamental_matrix.rs:17-69) with pixel coordinates of ~1e3 (design-matrix
    entries of ~1e6): the model is the right singular vector of the SMALLEST OF THE EIGHT singular values of the 8x9
    system (numpy's f64 SVD as the judge), unit length, and `None` when fewer than 8 singular values exceed epsilon."""
    rng = np.random.default_rng(5)
    for trial in range(20):
        k0 = np.zeros(8, amd.KEYPOINT_DTYPE)
        k1 = np.zeros(8, amd.KEYPOINT_DTYPE)
        k0["x"], k0["y"] = rng.uniform(0, 2000, 8), rng.uniform(0, 1500, 8)
        k1["x"], k1["y"] = k0["x"] + rng.uniform(-30, 30, 8), k0["y"] + rng.uniform(-30, 30, 8)
        m = np.zeros(8, amd.MATCH_DTYPE)
        m["index_0"] = m["index_1"] = np.arange(8)
        f = amd.estimate_fundamental_matrix(k0, k1, m, 0.05)
        x0, y0, x1, y1 = (k0["x"].astype(np.float32), k0["y"].astype(np.float32), k1["x"].astype(np.float32),
                          k1["y"].astype(np.float32))
        a = np.stack([x0 * x1, x0 * y1, x0, y0 * x1, y0 * y1, y0, x1, y1, np.ones(8, np.float32)], axis=1).astype(np.float64)
        _, sv, vt = np.linalg.svd(a, full_matrices=False)
        assert f is not None and sv[7] > 0.05
        v = np.array([f[0, 0], f[1, 0], f[2, 0], f[0, 1], f[1, 1], f[2, 1], f[0, 2], f[1, 2], f[2, 2]], np.float64)
        assert abs(np.linalg.norm(v) - 1.0) < 1e-5
        assert abs(abs(float(v @ vt[7])) - 1.0) < 1e-4, trial          # same direction up to sign
        assert abs(np.linalg.norm(a @ v) - sv[7]) < 1e-3 * max(1.0, sv[7])
    # degenerate: all correspondences identical -> rank 1 -> None
    k0["x"] = k0["y"] = k1["x"] = k1["y"] = 10.0
    assert amd.estimate_fundamental_matrix(k0, k1, m, 0.05) is None


def test_rust_shim_covers_the_reference_api():
    """The Rust shim cannot be compiled here (no rustc): tools/check_shim.py compares its public items — module paths,
    parameter names and types, return types, public struct fields, trait methods — with the reference crate's (read from
    the reference sources when present, else from the committed snapshot) and its extern "C" block with the header."""
    import subprocess, sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_shim.py")], capture_output=True, text=True)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "reference public items: 38" in p.stdout
    # every name the reference's own callers import (akaze-util/src/bin/*.rs, akaze/tests/integration-test.rs:9-11)
    import json
    api = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_api.json")))
    for name in ("extract_features", "match_features", "types::evolution::Config", "types::evolution::write_evolutions",
                 "types::keypoint::draw_keypoints_to_image", "types::feature_match::draw_matches",
                 "types::feature_match::Match", "types::keypoint::Keypoint", "types::keypoint::Descriptor"):
        assert name in api, name


def test_march_bands_tile_the_rows(amd):
    """The row bands of the column-march planners (host code; no GPU): for many heights, widths and batch sizes the
    bands are non-empty, ascending, disjoint and cover exactly the interior rows -- [S, h-S) for the detector / blur
    march, [1, h-1) for the level march with its shorter first and last band."""
    import ctypes as C
    lib = amd.lib()
    rows = (C.c_int32 * 512)()
    nb = C.c_uint32()
    checked = 0
    for kind, halves in ((0, (1, 2, 3, 4)), (1, (1,))):
        for S in halves:
            for h in (16, 17, 33, 64, 65, 100, 135, 270, 540, 1080, 1081, 2160, 4099):
                for w, n in ((16, 1), (480, 1), (481, 3), (1920, 1), (1920, 32), (960, 32), (3840, 8), (240, 256)):
                    if h < 4 * S + 8:
                        continue
                    assert lib.akz_debug_march_bands(kind, w, h, n, S, rows, 256, C.byref(nb)) == 0
                    k = nb.value
                    assert 1 <= k <= 256, (kind, S, w, h, n, k)
                    lo, hi = (S, h - S) if kind == 0 else (1, h - 1)
                    at = lo
                    for b in range(k):
                        cs, ce = rows[2 * b], rows[2 * b + 1]
                        assert cs == at and ce > cs, (kind, S, w, h, n, b, cs, ce, at)
                        at = ce
                    assert at == hi, (kind, S, w, h, n, at, hi)
                    checked += 1
    assert checked > 300


def test_placement_verdict_on_recorded_timings(amd):
    """The stream-placement probe's decision (akz_api.cpp, akz::placement_verdict) on the timings recorded in
    profiles/r04_queue_probe.txt: bit 0 = one hardware queue, bit 1 = one command-processor pipe, bit 2 = ambiguous (the
    probe repeats such a measurement and lets the shortest decide)."""
    v = amd.lib().akz_debug_placement_verdict
    spin = 0.120
    # one 120 us spin on each of two streams: 123-158 us side by side, 244-245 us on one hardware queue
    for t in (0.123, 0.133, 0.143, 0.157):
        assert v(t, -1.0, 0.126, spin) & 1 == 0, t
    for t in (0.244, 0.245, 0.300):
        assert v(t, -1.0, 0.126, spin) & 1 == 1, t
    assert v(0.180, -1.0, 0.126, spin) & 4 and v(0.200, -1.0, 0.126, spin) & 5 == 5   # near the threshold: ambiguous either way
    assert v(0.130, -1.0, 0.126, spin) & 4 == 0 and v(0.245, -1.0, 0.126, spin) & 4 == 0
    # 40 tiny kernels on each stream: 78-161 us alone, 198-315 us on different pipes, 913-1149 us on one pipe
    alone = 0.126
    for t in (0.198, 0.238, 0.292, 0.315):
        assert v(0.0, t, alone, spin) & 2 == 0 and v(0.0, t, alone, spin) & 4 == 0, t
    for t in (0.913, 0.997, 1.048, 1.149):
        assert v(0.0, t, alone, spin) == 2, t
    # 634 us -- one reading of that table, a pair on different pipes -- says "shared" once and is marked ambiguous: the probe
    # measures again, and the next (shortest) reading of such a pair, ~0.25 ms, clears it
    assert v(0.0, 0.634, alone, spin) == 6
    assert v(0.0, min(0.634, 0.245), alone, spin) == 0


def test_remove_outliers_pool_survives_fork(amd):
    """akz_remove_outliers' process-wide pool of trial threads (akz_ransac.cpp) after fork(): the child inherits the pool
    object but none of its threads -- it must notice (pid), start its own, give the parent's result for the same random
    source, and exit cleanly (round-4 advice)."""
    rng = np.random.default_rng(5)
    n = 3000
    k0 = np.zeros(n, amd.KEYPOINT_DTYPE)
    k1 = np.zeros(n, amd.KEYPOINT_DTYPE)
    k0["x"], k0["y"] = rng.uniform(0, 1900, n), rng.uniform(0, 1000, n)
    k1["x"], k1["y"] = k0["x"] + 17 + rng.normal(0, 0.4, n), k0["y"] + 9 + rng.normal(0, 0.4, n)
    m = np.zeros(n, amd.MATCH_DTYPE)
    m["index_0"] = m["index_1"] = np.arange(n)
    amd.random_seed(42, 69)
    first = amd.remove_outliers(k0, k1, m, 1000, 0.05, 3.0)   # grows the pool in THIS process
    assert 8 <= len(first) <= n
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:
        rc = 1
        try:
            os.close(r)
            amd.random_seed(42, 69)
            again = amd.remove_outliers(k0, k1, m, 1000, 0.05, 3.0)
            rc = 0 if np.array_equal(again, first) else 2
            os.write(w, b"k")
        finally:
            os._exit(rc)
    os.close(w)
    import select
    ready, _, _ = select.select([r], [], [], 60)
    if not ready:
        os.kill(pid, 9)
    _, status = os.waitpid(pid, 0)
    assert ready and os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    amd.random_seed(42, 69)
    assert np.array_equal(amd.remove_outliers(k0, k1, m, 1000, 0.05, 3.0), first)   # the parent's pool is untouched
