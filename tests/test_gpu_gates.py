"""The gate matrix, enumerated (csrc/akz_gates.hpp; akz_debug_gates): every gate picks between two EQUIVALENT forms, so each
side of each gate is forced on each BASELINE geometry through both entry points -- the synchronous call and the begin /
finish interface -- and every result must be the one the default settings give, which is held to the oracle on the same
geometries (planes included) by tests/test_gpu_headline.py and, here, once more for the smallest of them."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# BASELINE configs[1..4] as one-GPU jobs: (name, w, h, frames, Config overrides)
GEOMETRIES = [
    ("c2_lone_1080p", 1920, 1080, 1, {}),
    ("c3_pair_4k", 3840, 2160, 2, {}),
    ("c4_batch_1080p", 1920, 1080, 4, {}),
    ("c5_4k_5x5", 3840, 2160, 1, dict(num_sublevels=5, max_octave_evolution=5)),
]

# (what is forced, how, how it is undone): one side of a gate each; the job gates take both sides explicitly
FORCED = [
    ("default", lambda c: None, lambda c: None),
    ("job gates: batch path for every job", lambda c: c.debug_set_schedule(4, 1), lambda c: c.debug_set_schedule(4, 0)),
    ("job gates: lone path for every job", lambda c: c.debug_set_schedule(4, 2_000_000), lambda c: c.debug_set_schedule(4, 0)),
    ("march_px / level_march_px: marches wherever supported", lambda c: (c.set_prep_mode(3), c.set_detector_mode(5)),
     lambda c: (c.set_prep_mode(2), c.set_detector_mode(2))),
    ("march_px: tiled detector, stream_px: streaming preparation", lambda c: (c.set_prep_mode(1), c.set_detector_mode(4)),
     lambda c: (c.set_prep_mode(2), c.set_detector_mode(2))),
    ("stream_px: tiled preparation, tiled detector pair", lambda c: (c.set_prep_mode(0), c.set_detector_mode(0)),
     lambda c: (c.set_prep_mode(2), c.set_detector_mode(2))),
    ("fed: one launch per step", lambda c: c.set_fed_mode(0), lambda c: c.set_fed_mode(2)),
    ("tiled family, level 0 as separate launches and every preparation a launch of its own",
     lambda c: (c.set_prep_mode(0), c.debug_set_schedule(10, 1), c.debug_set_schedule(6, 1)),
     lambda c: (c.set_prep_mode(2), c.debug_set_schedule(10, 0), c.debug_set_schedule(6, 0))),
    ("results by three copies behind the descriptor kernel", lambda c: c.debug_set_schedule(9, 1), lambda c: c.debug_set_schedule(9, 0)),
    ("column marches cut into 16-row bands", lambda c: (c.debug_set_schedule(7, 16), c.debug_set_schedule(8, 16)),
     lambda c: (c.debug_set_schedule(7, 0), c.debug_set_schedule(8, 0))),
    ("few_host_threads: selection and sort on the device", lambda c: (c.debug_set_select(2), c.debug_set_host_sort(False)),
     lambda c: (c.debug_set_select(None), c.debug_set_host_sort(None))),
    ("select_device_px: neighbour lists + host selection", lambda c: c.debug_set_select(1), lambda c: c.debug_set_select(None)),
    ("few_host_threads: host sort + host grids", lambda c: (c.debug_set_select(0), c.debug_set_host_sort(True)),
     lambda c: (c.debug_set_select(None), c.debug_set_host_sort(None))),
]


def _digest(res, n):
    h = hashlib.sha256()
    for img in range(n):
        h.update(res.keypoints(img).tobytes())
        h.update(res.descriptors(img).tobytes())
        h.update(np.float64(res.contrast(img)).tobytes())
    nl = res.counts(0)[0]
    for lvl, name in ((1, "Lt"), (2, "Lflow"), (3, "Ldet"), (nl // 2, "Lsmooth"), (nl - 1, "Lt"), (nl - 1, "Lxy"), (nl - 2, "Lstep")):
        h.update(np.ascontiguousarray(res.plane(lvl, name, n - 1)).tobytes())
    return h.hexdigest()


def test_gate_table_is_complete_and_named(amd):
    rows = amd.gates()
    names = [r["name"] for r in rows]
    assert len(names) == len(set(names)) >= 12
    for want in ("big_px_sync", "big_px_async", "march_px", "level_march_px", "stream_px", "few_host_threads", "select_device_px",
                 "sort_buckets", "merge_compact_min_rows", "merge_compact_max_rows"):
        assert want in names
    assert all(r["value"] > 0 and r["unit"] and len(r["meaning"]) > 20 for r in rows)
    by = {r["name"]: r["value"] for r in rows}
    assert by["big_px_async"] <= by["big_px_sync"] <= by["march_px"]  # a begin / finish job takes the batch path no later than a call


@pytest.mark.parametrize("geo", GEOMETRIES, ids=[g[0] for g in GEOMETRIES])
def test_every_gate_side_on_every_baseline_geometry(ctx, amd, ref, geo):
    import torch
    name, w, h, n, kw = geo
    cfg = amd.Config(**kw)
    frames = np.stack([amd.synth_frame(w, h, 60 + i) for i in range(n)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    want = None
    try:
        for label, force, undo in FORCED:
            force(ctx)
            try:
                sync = ctx.extract_features(d, cfg, keep_all_planes=True)
                got_sync = _digest(sync, n)
                if want is None:
                    want = got_sync
                    if name == "c2_lone_1080p":  # the default on this geometry against the oracle, every keypoint field and byte
                        rf = ref.extract(frames[0], ref.default_config(**kw), threads=8)
                        kp, rk = sync.keypoints(0), rf.keypoints()
                        assert len(kp) == len(rk) > 1000 and all(np.array_equal(kp[f], rk[f]) for f in ("x", "y", "response", "angle"))
                        assert np.array_equal(sync.descriptors(0), rf.descriptors())
                        rf.close()
                sync.close()
                asyn = ctx.extract_begin(d, cfg, keep_all_planes=True).finish()
                got_asyn = _digest(asyn, n)
                asyn.close()
                assert got_sync == want, (name, label, "synchronous call")
                assert got_asyn == want, (name, label, "begin / finish")
            finally:
                undo(ctx)
    finally:
        for _, _, undo in FORCED:
            undo(ctx)


def test_matcher_gates_both_sides(ctx, amd, ref):
    """merge_compact_min_rows / premerge_chunks / the three matcher kernels: query sets either side of 2 048 rows, few and many
    train chunks -- every combination equals the oracle's descriptor_match"""
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (2600, 61), dtype=np.uint8)
    noisy = base ^ (rng.random((2600, 61)) < 0.02).astype(np.uint8)  # near-duplicates: plenty of matches pass the ratio test
    try:
        for n0 in (1500, 2047, 2048, 2600):
            want = ref.descriptor_match(base[:n0], noisy, 10000, 0.86)
            assert len(want) > n0 // 2
            for mode in (2, 1, 0):
                ctx.set_match_mode(mode)
                for chunks in (0, 2, 7):
                    ctx.debug_set_match_chunks(chunks, 0)
                    got = ctx.descriptor_match(base[:n0], noisy, 10000, 0.86)
                    assert np.array_equal(got, want), (n0, mode, chunks)
    finally:
        ctx.set_match_mode(2)
        ctx.debug_set_match_chunks(0, 0)


@pytest.mark.parametrize("w,h", [(640, 480), (1280, 720), (500, 400)])
def test_lean_jobs_of_the_begin_finish_interface(ctx, amd, ref, w, h):
    """big_px_async: a begun job takes the batch path from 0.3 Mpx on (a synchronous call from 1.4 Mpx): both sides of it, lean
    planes, pipelined two deep, against the oracle and against the other entry point / plane set."""
    import torch
    frames = [amd.synth_frame(w, h, 90 + i) for i in range(3)]
    dev = [torch.from_numpy(f[None]).cuda() for f in frames]
    torch.cuda.synchronize()
    jobs = [ctx.extract_begin(dev[0], keep_all_planes=False), ctx.extract_begin(dev[1], keep_all_planes=False)]
    res = [jobs[0].finish()]
    jobs.append(ctx.extract_begin(dev[2], keep_all_planes=False))
    res += [jobs[1].finish(), jobs[2].finish()]
    for i in range(3):
        rf = ref.extract(frames[i])
        kp, rk = res[i].keypoints(0), rf.keypoints()
        assert len(kp) == len(rk) > 50
        assert all(np.array_equal(kp[f], rk[f]) for f in ("x", "y", "response", "size", "octave", "class_id", "angle"))
        assert np.array_equal(res[i].descriptors(0), rf.descriptors())
        assert float(res[i].contrast(0)) == rf.contrast
        sync = ctx.extract_features(dev[i], keep_all_planes=False)
        full = ctx.extract_begin(dev[i]).finish()
        for other in (sync, full):
            assert other.keypoints(0).tobytes() == kp.tobytes() and other.descriptors(0).tobytes() == res[i].descriptors(0).tobytes()
            other.close()
        rf.close()
        res[i].close()


def test_calibrated_job_gates_change_nothing_but_the_gates(ctx, amd):
    """akz_ctx_calibrate_gates: the two job gates re-derived from timings on this machine -- values among the measured sizes,
    and results before / after identical."""
    import torch
    d = torch.from_numpy(amd.synth_frame(2016, 1512, 3)).cuda()
    before = ctx.extract_features(d)
    dig = _digest(before, 1)
    before.close()
    try:
        sync_px, async_px, ms = ctx.calibrate_gates()
        sizes = [1280 * 720, 1600 * 900, 1920 * 1080, 2688 * 1512, 3840 * 2160, 3840 * 2160 + 1]
        assert sync_px in sizes and async_px in sizes
        assert all(0.05 < v < 50 for row in ms for v in row), ms
        after = ctx.extract_features(d)
        assert _digest(after, 1) == dig
        after.close()
        j = ctx.extract_begin(d, amd.Config()).finish()
        assert _digest(j, 1) == dig
        j.close()
    finally:
        ctx.debug_set_schedule(4, 0)   # back to the compiled-in gates for the rest of the session
