"""GPU end-to-end parity of extract_features / descriptor_match against the CPU oracle and the
committed golden vectors.  Bar: every EvolutionStep plane, keypoint field, descriptor byte and
match record identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PLANES = ["Lt", "Lsmooth", "Lx", "Ly", "Lxx", "Lyy", "Lxy", "Lflow", "Lstep", "Ldet"]
KP_FIELDS = ("x", "y", "response", "size", "octave", "class_id", "angle")


def assert_same_result(res, rf, planes=True, img=0, equal_nan=False):
    nl, nk, nb = res.counts(img)
    assert nl == rf.num_levels and nb == rf.desc_bytes
    assert res.contrast(img) == rf.contrast or (equal_nan and np.isnan(res.contrast(img)) and np.isnan(rf.contrast))
    if planes:
        for lvl in range(nl):
            info, rinfo = res.level_info(lvl), rf.level_info(lvl)
            for f in ("etime", "esigma", "octave", "sublevel", "sigma_size", "w", "h"):
                assert info[f] == rinfo[f], (lvl, f)
            assert info["tau"].tobytes() == rinfo["tau"].tobytes()
            for pl in PLANES:
                a, b = res.plane(lvl, pl, img), rf.plane(lvl, pl)
                assert a.shape == b.shape, (lvl, pl, a.shape, b.shape)
                if a.size and not np.array_equal(a, b, equal_nan=equal_nan):
                    bad = np.argwhere(a != b)
                    raise AssertionError(f"level {lvl} plane {pl}: {len(bad)} px differ, first {bad[0]}: "
                                         f"{a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}")
    kp, rk = res.keypoints(img), rf.keypoints()
    assert nk == rf.num_keypoints == len(kp)
    for f in KP_FIELDS:
        assert np.array_equal(kp[f], rk[f]), f
    assert np.array_equal(res.descriptors(img), rf.descriptors())


@pytest.mark.parametrize("w,h,idx", [(320, 240, 0), (640, 480, 1), (517, 389, 2)])
def test_extract_matches_oracle_all_planes(ctx, amd, ref, w, h, idx):
    frame = amd.synth_frame(w, h, idx)
    res = ctx.extract_features(frame)
    rf = ref.extract(frame)
    assert rf.num_keypoints > 0
    assert_same_result(res, rf)


@pytest.mark.parametrize("w,h", [(11, 11), (12, 40), (33, 17), (81, 41), (159, 79), (160, 80), (161, 81), (2000, 24),
                                 (24, 1500), (70, 50), (162, 83), (166, 82), (67, 133)])
def test_extract_edge_sizes(ctx, amd, ref, w, h):
    """The smallest accepted frame (11 x 11: the detector Scharr of sigma 4 needs 2*4+3), frames smaller than one
    tile, strips, and the sizes either side of the second octave's admission rule (evolution.rs:138-149: 160 x 80)."""
    frame = amd.synth_frame(w, h, 3)
    res = ctx.extract_features(frame)
    rf = ref.extract(frame)
    assert res.counts(0)[0] == rf.num_levels == (8 if w >= 160 and h >= 80 else 4)
    assert_same_result(res, rf)


@pytest.mark.parametrize("w,h,n", [(163, 81, 3), (130, 66, 4), (321, 243, 2)])
def test_extract_small_batches_of_odd_frames(ctx, amd, ref, w, h, n):
    """Several frames per job whose planes do not start on 16-byte boundaries (w * h odd or 2 mod 4: the elementwise
    histogram and flow kernels of the level-0 launches take their scalar form) and whose last column is not the last of a
    group of four (k_fed_own's border substitution at every place of a group): every image equals the oracle's, planes included."""
    import torch
    frames = np.stack([amd.synth_frame(w, h, 60 + i) for i in range(n)])
    res = ctx.extract_features(torch.from_numpy(frames).cuda())
    for i in range(n):
        rf = ref.extract(frames[i])
        assert_same_result(res, rf, img=i)
        rf.close()


@pytest.mark.parametrize("w,h,block", [(200, 120, 8), (163, 81, 5), (640, 360, 16), (322, 200, 7)])
def test_diffusion_planes_bit_for_bit_on_blocky_frames(ctx, amd, ref, w, h, block):
    """Frames of constant blocks: most fluxes of the diffusion are exact zeros, at the image's border too -- where k_fed_own
    replaces the missing term of the reference's shorter expressions by a signed zero (nonlinear_diffusion.rs:84-137).  Lt and
    Lstep of every level must equal the oracle's BIT FOR BIT (np.array_equal takes -0.0 for +0.0: compared as words here)."""
    rng = np.random.default_rng(w * 31 + h)
    coarse = rng.integers(0, 256, ((h + block - 1) // block, (w + block - 1) // block), dtype=np.uint8)
    coarse[rng.random(coarse.shape) < 0.5] = 0  # (half of the blocks black: exact zeros in the planes themselves)
    frame = np.ascontiguousarray(np.kron(coarse, np.ones((block, block), np.uint8))[:h, :w])
    res, rf = ctx.extract_features(frame), ref.extract(frame)
    nl = res.counts(0)[0]
    assert nl == rf.num_levels
    for lvl in range(nl):
        for pl in ("Lt", "Lstep", "Lflow", "Lsmooth"):
            a, b = np.ascontiguousarray(res.plane(lvl, pl, 0)), np.ascontiguousarray(rf.plane(lvl, pl))
            if a.size == 0:
                continue
            nan = np.isnan(a)
            assert np.array_equal(nan, np.isnan(b)), (lvl, pl)
            wa, wb = a.view(np.uint32)[~nan], b.view(np.uint32)[~nan]
            if not np.array_equal(wa, wb):
                bad = np.flatnonzero(wa != wb)
                raise AssertionError(f"level {lvl} plane {pl}: {len(bad)} words differ, first {hex(wa[bad[0]])} vs {hex(wb[bad[0]])}")
    assert_same_result(res, rf)
    rf.close()


def test_extract_flat_and_noise_frames(ctx, amd, ref):
    """A constant frame (no gradient anywhere: the diffusion planes turn NaN in the reference's arithmetic and must
    do so here, NaN payloads aside; no extrema) and uniform noise (every NMS neighbourhood busy: 1131 keypoints on
    320 x 240)."""
    flat = np.full((120, 200), 77, np.uint8)
    res, rf = ctx.extract_features(flat), ref.extract(flat)
    assert rf.num_keypoints == 0
    assert_same_result(res, rf, equal_nan=True)
    noise = np.random.default_rng(1).integers(0, 256, (240, 320), dtype=np.uint8)
    res, rf = ctx.extract_features(noise), ref.extract(noise)
    assert rf.num_keypoints > 1000
    assert_same_result(res, rf)


def test_extract_f32_input_and_lean_planes(ctx, amd, ref):
    frame = (amd.synth_frame(400, 300, 5).astype(np.float32) * np.float32(1.0)) / np.float32(255.0)
    res = ctx.extract_features(frame, keep_all_planes=False)
    rf = ref.extract(frame)
    assert_same_result(res, rf, planes=False)
    # planes that are not kept are recomputed on fetch, kept ones are still exact
    assert np.array_equal(res.plane(3, "Lxx"), rf.plane(3, "Lxx")) and np.array_equal(res.plane(3, "Lstep"), rf.plane(3, "Lstep"))
    assert np.array_equal(res.plane(3, "Ldet"), rf.plane(3, "Ldet"))
    assert res.plane(0, "Lstep").size == 0 and rf.plane(0, "Lstep").size == 0
    assert res.plane(0, "Lflow").size == 0 and rf.plane(0, "Lflow").size == 0


def test_extract_device_batch(ctx, amd, ref):
    import torch
    frames = np.stack([amd.synth_frame(480, 270, i) for i in range(3)])
    res = ctx.extract_features(torch.from_numpy(frames).cuda())
    assert res.num_images == 3
    for i in range(3):
        assert_same_result(res, ref.extract(frames[i]), planes=(i == 1), img=i)


def test_extract_nondefault_config(ctx, amd, ref):
    frame = amd.synth_frame(640, 360, 7)
    for kw in (dict(num_sublevels=5, max_octave_evolution=5), dict(descriptor_channels=1),
               dict(descriptor_channels=2), dict(detector_threshold=0.0005, num_sublevels=3)):
        res = ctx.extract_features(frame, amd.Config(**kw))
        rf = ref.extract(frame, ref.default_config(**kw))
        assert_same_result(res, rf, planes=False)


def test_golden_vectors(ctx, amd):
    g = np.load(os.path.join(GOLDEN, "synthetic_small.npz"))
    descs = {}
    for name in ("a", "b"):
        res = ctx.extract_features(g[f"{name}_frame"])
        assert np.array_equal(res.keypoints().view(np.uint8), g[f"{name}_keypoints"])
        assert np.array_equal(res.descriptors(), g[f"{name}_descriptors"])
        assert res.contrast() == float(g[f"{name}_contrast"])
        descs[name] = res.descriptors()
    m = ctx.descriptor_match(descs["a"], descs["b"], 10000, 0.86)
    assert np.array_equal(m.view(np.uint8), g["matches_ab"])


def test_extract_and_match_pair(ctx, amd, ref):
    """Shape of the reference's integration test (akaze/tests/integration-test.rs:73-93) on a synthetic
    pair: extract both views, descriptor_match with ratio 0.86, compare with the oracle."""
    f0 = amd.synth_frame(960, 540, 11)
    f1 = amd.synth_frame(960, 540, 11, shift=(17, 9))
    r0, r1 = ctx.extract_features(f0), ctx.extract_features(f1)
    q0, q1 = ref.extract(f0), ref.extract(f1)
    assert_same_result(r0, q0, planes=False)
    assert_same_result(r1, q1, planes=False)
    got = amd.match_features(r0.keypoints(), r0.descriptors(), r1.keypoints(), r1.descriptors(), 0.86, ctx=ctx)
    exp = ref.descriptor_match(q0.descriptors(), q1.descriptors(), 10000, 0.86)
    assert len(exp) > 20 and np.array_equal(got, exp)


def test_errors(ctx, amd):
    with pytest.raises(amd.AkazeError) as e:
        ctx.extract_features(np.zeros((8, 8), np.uint8))
    assert e.value.status == -4
    with pytest.raises(amd.AkazeError):
        ctx.extract_features(amd.synth_frame(320, 240, 0), amd.Config(descriptor_channels=4))
    with pytest.raises(amd.AkazeError) as e:
        ctx.extract_features(amd.synth_frame(320, 240, 0), amd.Config(num_sublevels=16))
    assert e.value.status in (-1, -6)


def test_begin_finish_pipelined_matches_oracle(ctx, amd, ref):
    """Two extractions in flight on one context (begin B before finishing A) give the same results as
    the synchronous call; a third begin is accepted, a fourth is refused until one is finished."""
    import torch
    fa = np.stack([amd.synth_frame(480, 270, i) for i in range(2)])
    fb = np.stack([amd.synth_frame(480, 270, 10 + i) for i in range(3)])
    ctx.set_eager_finish(False)  # (with the finish half on the context's own thread -- the default -- a job gives its slot
    try:                         #  back as soon as that thread has fetched its candidates, whenever that is)
        ja = ctx.extract_begin(torch.from_numpy(fa).cuda())
        jb = ctx.extract_begin(torch.from_numpy(fb).cuda())
        jc = ctx.extract_begin(torch.from_numpy(fa).cuda())
        with pytest.raises(amd.AkazeError):
            ctx.extract_begin(torch.from_numpy(fa).cuda())
        ra = ja.finish()
        rb = jb.finish()
        del jc  # abandoned without finishing
    finally:
        ctx.set_eager_finish(True)
    # the same three jobs with the finish half on the context's own thread
    ea = ctx.extract_begin(torch.from_numpy(fa).cuda())
    eb = ctx.extract_begin(torch.from_numpy(fb).cuda())
    ec = ctx.extract_begin(torch.from_numpy(fa).cuda())
    rea, reb = ea.finish(), eb.finish()
    del ec
    for i in range(2):
        assert rea.keypoints(i).tobytes() == ra.keypoints(i).tobytes() and rea.descriptors(i).tobytes() == ra.descriptors(i).tobytes()
    for i in range(3):
        assert reb.keypoints(i).tobytes() == rb.keypoints(i).tobytes() and reb.descriptors(i).tobytes() == rb.descriptors(i).tobytes()
    for i in range(2):
        assert_same_result(ra, ref.extract(fa[i]), planes=(i == 0), img=i)
    for i in range(3):
        assert_same_result(rb, ref.extract(fb[i]), planes=False, img=i)
    # the context is still usable and its slots are free again
    r = ctx.extract_features(fa[0])
    assert_same_result(r, ref.extract(fa[0]), planes=False)


def test_lanes_stream_of_single_frames(amd, ref):
    """akz_ctx_set_lanes: single frames dealt to three child contexts, several jobs in flight, results identical to the
    oracle in every plane; a batch above the lane threshold runs on the context itself; lanes can be changed between
    jobs and the context is destroyed with results of its lanes still alive."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    c.set_lanes(3)
    frames = [amd.synth_frame(640, 360, 40 + i) for i in range(7)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    jobs = [c.extract_begin(d) for d in dev[:5]]          # five jobs in flight over three lanes
    res = [j.finish() for j in jobs]
    jobs = [c.extract_begin(d) for d in dev[5:]]
    res += [j.finish() for j in jobs]
    for f, r in zip(frames, res):
        assert_same_result(r, ref.extract(f), planes=True)
    big = np.stack([amd.synth_frame(1920, 1080, i) for i in range(5)])  # 10 Mpx: not dealt to a lane
    rb = c.extract_begin(torch.from_numpy(big).cuda()).finish()
    assert_same_result(rb, ref.extract(big[4], threads=8), planes=False, img=4)
    c.set_lanes(1)
    r1 = c.extract_begin(dev[0]).finish()
    assert_same_result(r1, ref.extract(frames[0]), planes=False)
    c.set_lanes(2)
    r2 = c.extract_begin(dev[1]).finish()
    c.close()                                             # results of lanes outlive the context
    assert r2.keypoints(0).tobytes() == res[1].keypoints(0).tobytes()
    assert r2.descriptors(0).tobytes() == res[1].descriptors(0).tobytes()


def test_eager_finish_on_lanes(amd, ref):
    """akz_ctx_set_eager_finish: the finish half of jobs on lanes runs on the lanes' own threads.  A stream of frames with
    several jobs in flight (more than one per lane), frames of different sizes, a job that is abandoned, an overflowing
    candidate list (the finish half's retry path on the thread), a result freed while its lane is busy with a later job,
    other calls on the context in between (they wait for the threads), switching the mode off again and destroying the
    context with results alive: every plane, keypoint and descriptor byte against the oracle."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    c.set_lanes(3)
    c.set_eager_finish(True)
    sizes = [(640, 360), (505, 393), (640, 360), (161, 81), (640, 360), (800, 450), (640, 360), (480, 270)]
    frames = [amd.synth_frame(w, h, 70 + i) for i, (w, h) in enumerate(sizes)]
    dev = [torch.from_numpy(f).cuda() for f in frames]
    refs = [ref.extract(f) for f in frames]
    res = [None] * len(frames)
    pending = []
    for i, d in enumerate(dev):                            # five jobs in flight over three lanes
        pending.append((i, c.extract_begin(d)))
        if len(pending) == 5:
            k, j = pending.pop(0)
            res[k] = j.finish()
    for k, j in pending:
        res[k] = j.finish()
    for r, rf in zip(res, refs):
        assert_same_result(r, rf, planes=True)
    # a job nobody finishes, then the same frame again
    c.extract_begin(dev[1]).abandon()
    j_a, j_b = c.extract_begin(dev[1]), c.extract_begin(dev[5])
    c.synchronize()                                        # a call on the context between begin and finish
    assert_same_result(j_b.finish(), refs[5], planes=False)
    assert_same_result(j_a.finish(), refs[1], planes=False)
    # candidate overflow: the retry of the extrema pass runs on the lane's thread
    c.set_lanes(1)
    c.set_candidate_hint(16)
    c.set_lanes(2)
    ro = [c.extract_begin(dev[0]), c.extract_begin(dev[5])]
    assert_same_result(ro[1].finish(), refs[5], planes=False)
    assert_same_result(ro[0].finish(), refs[0], planes=False)
    # results freed while their lanes work on later jobs
    hold = [c.extract_begin(d) for d in dev[:4]]
    first = [hold[0].finish(), hold[1].finish()]
    more = [c.extract_begin(d) for d in dev[4:6]]
    first[0].close(), first[1].close()
    for j, k in zip(hold[2:] + more, (2, 3, 4, 5)):
        assert_same_result(j.finish(), refs[k], planes=False)
    c.set_eager_finish(False)
    assert_same_result(c.extract_begin(dev[7]).finish(), refs[7], planes=False)
    c.set_eager_finish(True)
    last = c.extract_begin(dev[6]).finish()
    unfinished = c.extract_begin(dev[2])                  # still owned by its lane's thread when the context goes away
    c.close()
    assert last.keypoints(0).tobytes() == res[6].keypoints(0).tobytes()
    assert last.descriptors(0).tobytes() == res[6].descriptors(0).tobytes()
    unfinished.abandon()


def test_eager_finish_soak_short():
    """tools/eager_soak.py for a few seconds: lone frames of six sizes over 2-4 lanes with eager finish, random abandons,
    late frees, lane re-sizing; every result against the synchronous extraction of the same frame (a 90 s run checks
    ~48 000 frames)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "eager_soak.py"), "4"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "no mismatch" in p.stdout


def test_baseline_c2_1080p_frame(ctx, amd, ref):
    """BASELINE.json configs[1]: one 1920x1080 synthetic frame, 4 octaves x 4 sublevels."""
    frame = amd.synth_frame(1920, 1080, 0)
    res = ctx.extract_features(frame)
    rf = ref.extract(frame, threads=8)
    assert rf.num_levels == 16 and rf.num_keypoints > 1000
    assert_same_result(res, rf, planes=False)
    for lvl, pl in ((0, "Lt"), (3, "Lflow"), (7, "Ldet"), (15, "Lt"), (15, "Ldet")):
        assert np.array_equal(res.plane(lvl, pl), rf.plane(lvl, pl)), (lvl, pl)


def test_baseline_c3_4k_pair_extract_and_match(ctx, amd, ref):
    """BASELINE.json configs[2]: 3840x2160 synthetic pair, extract_features + Hamming match_features;
    descriptor bytes and match index pairs identical to the oracle."""
    f0 = amd.synth_frame(3840, 2160, 0)
    f1 = amd.synth_frame(3840, 2160, 0, shift=(17, 9))
    r0, r1 = ctx.extract_features(f0, keep_all_planes=False), ctx.extract_features(f1, keep_all_planes=False)
    q0, q1 = ref.extract(f0, threads=8), ref.extract(f1, threads=8)
    assert q0.num_keypoints > 4000
    assert_same_result(r0, q0, planes=False)
    assert_same_result(r1, q1, planes=False)
    got = ctx.descriptor_match(r0.descriptors(), r1.descriptors(), 10000, 0.86)
    exp = ref.descriptor_match(q0.descriptors(), q1.descriptors(), 10000, 0.86)
    assert len(exp) > 500 and np.array_equal(got, exp)
    # size-independent properties: matching a set against itself pairs every descriptor with itself at
    # distance 0 unless it has an exact duplicate (then the ratio test rejects it)
    d0 = r0.descriptors()
    self_m = ctx.descriptor_match(d0, d0, 10000, 0.86)
    assert np.all(self_m["index_0"] == self_m["index_1"]) and np.all(self_m["distance"] == 0)
    assert len(self_m) >= len(d0) - 2 * (len(d0) - len(np.unique(d0, axis=0)))


def test_match_features_full_dropin(ctx, amd, ref):
    """akaze::match_features with its reference signature (lib.rs:252-260): GPU descriptor_match followed by
    the host RANSAC filter, against the oracle's descriptor_match + remove_outliers."""
    f0 = amd.synth_frame(960, 540, 11)
    f1 = amd.synth_frame(960, 540, 11, shift=(17, 9))
    r0, r1 = ctx.extract_features(f0, keep_all_planes=False), ctx.extract_features(f1, keep_all_planes=False)
    q0, q1 = ref.extract(f0), ref.extract(f1)
    amd.random_seed(42, 69)   # product and oracle each keep one persistent random source per thread
    ref.random_seed(42, 69)
    got = amd.match_features(r0.keypoints(), r0.descriptors(), r1.keypoints(), r1.descriptors(), 0.86, 1000, 3.0,
                             ctx=ctx)
    raw = ref.descriptor_match(q0.descriptors(), q1.descriptors(), 10000, 0.86)
    exp = ref.remove_outliers(q0.keypoints(), q1.keypoints(), raw, 1000, 0.05, 3.0)
    assert len(raw) >= 8 and np.array_equal(got, exp)
    assert set(got["index_0"].tolist()) <= set(raw["index_0"].tolist())


def test_match_features_trials_on_device_equal_host_trials(ctx, amd):
    """match_features runs its RANSAC trials on the device (akz_fmatrix.hip: one workgroup per trial, the model from the same
    source as the host's, akz_fmatrix.hpp); remove_outliers runs them on host threads.  Same samples (the calling thread's
    random source), same models bit for bit, same winner: identical lists -- on a real pair, on a scene where many samples
    are rank-deficient (collinear points: no model) and with more trials than matches."""
    f0 = amd.synth_frame(1280, 720, 5)
    f1 = amd.synth_frame(1280, 720, 5, shift=(11, 6))
    r0, r1 = ctx.extract_features(f0, keep_all_planes=False), ctx.extract_features(f1, keep_all_planes=False)
    k0, d0, k1, d1 = r0.keypoints(), r0.descriptors(), r1.keypoints(), r1.descriptors()
    raw = ctx.descriptor_match(d0, d1, 10000, 0.86)
    assert len(raw) > 500
    for trials, eps in ((1000, 3.0), (4000, 0.5), (997, 10.0)):
        amd.random_seed(42, 69)
        host = amd.remove_outliers(k0, k1, raw, trials, 0.05, eps)
        amd.random_seed(42, 69)
        dev = amd.match_features(k0, d0, k1, d1, 0.86, trials, eps, ctx=ctx)
        assert np.array_equal(dev, host) and 8 <= len(dev) <= len(raw), (trials, eps)
    # collinear keypoints in image 0 (y = 2 x): the design matrix of most samples loses rank -> no model for those trials
    kc = k0.copy()
    kc["y"] = 2.0 * kc["x"]
    amd.random_seed(1, 2)
    host = amd.remove_outliers(kc, k1, raw, 2000, 0.05, 3.0)
    amd.random_seed(1, 2)
    dev = amd.match_features(kc, d0, k1, d1, 0.86, 2000, 3.0, ctx=ctx)
    assert np.array_equal(dev, host)


def test_baseline_c4_batch_is_independent_per_image(ctx, amd, ref):
    """BASELINE.json configs[3] (a batch of 1080p frames sharded one image per GPU slot): a frame's result does
    not depend on what else is in the batch or where it sits, and equals the oracle's for a sampled frame."""
    import torch
    frames = np.stack([amd.synth_frame(1920, 1080, 100 + i) for i in range(6)])
    batch = ctx.extract_features(torch.from_numpy(frames).cuda(), keep_all_planes=False)
    rev = ctx.extract_features(torch.from_numpy(frames[::-1].copy()).cuda(), keep_all_planes=False)
    for i in range(6):
        single = ctx.extract_features(frames[i], keep_all_planes=False)
        for other, j in ((batch, i), (rev, 5 - i)):
            assert single.counts(0) == other.counts(j)
            assert single.keypoints(0).tobytes() == other.keypoints(j).tobytes()
            assert np.array_equal(single.descriptors(0), other.descriptors(j))
            assert single.contrast(0) == other.contrast(j)
    assert_same_result(batch, ref.extract(frames[3], threads=8), planes=False, img=3)


def test_baseline_c5_4k_5x5_stream_all_pairs_match(ctx, amd, ref):
    """BASELINE.json configs[4] on one GPU: 3840x2160 frames, 5 octaves x 5 sublevels, full 486-bit M-LDB,
    then the all-pairs Hamming match over the gathered descriptor sets; everything identical to the oracle."""
    import torch
    kw = dict(num_sublevels=5, max_octave_evolution=5)
    frames = np.stack([amd.synth_frame(3840, 2160, 7, shift=(s, s // 2)) for s in (0, 12, 24)])
    res = ctx.extract_features(torch.from_numpy(frames).cuda(), amd.Config(**kw), keep_all_planes=False)
    refs = [ref.extract(f, ref.default_config(**kw), threads=8) for f in frames]
    for i, rf in enumerate(refs):
        assert rf.num_levels == 25 and rf.desc_bytes == 61 and rf.num_keypoints > 3000
        assert_same_result(res, rf, planes=False, img=i)
    n_matches = 0
    for i in range(3):
        for j in range(3):
            if i == j:
                continue
            got = ctx.descriptor_match(res.descriptors(i), res.descriptors(j), 10000, 0.86)
            exp = ref.descriptor_match(refs[i].descriptors(), refs[j].descriptors(), 10000, 0.86)
            assert np.array_equal(got, exp), (i, j)
            n_matches += len(exp)
    assert n_matches > 1000


def test_planes_not_kept_are_recomputed_on_fetch(ctx, amd, ref):
    """Without keep_all_planes Lxx / Lyy / Lxy / Lstep are not written by the extraction; fetching them recomputes
    them from the kept planes (second derivatives from Lsmooth, Lstep by repeating the level's diffusion, including
    the first level of an octave): every plane of every level equals the oracle's, for a frame of a batch."""
    import torch
    frames = np.stack([amd.synth_frame(517, 389, 30 + i) for i in range(2)])
    res = ctx.extract_features(torch.from_numpy(frames).cuda(), keep_all_planes=False)
    rf = ref.extract(frames[1])
    assert_same_result(res, rf, planes=True, img=1)
    cfg5 = dict(num_sublevels=5, max_octave_evolution=5)  # detector scales up to 5: the unfused derivative path
    res5 = ctx.extract_features(frames[0], amd.Config(**cfg5), keep_all_planes=False)
    assert_same_result(res5, ref.extract(frames[0], ref.default_config(**cfg5)), planes=True)


def test_candidate_list_overflow_is_retried(amd, ref):
    """A candidate list that is too small (16 entries per image here) overflows in the fused detector kernels; finish
    notices, enlarges the list and repeats the extrema pass on the stored Ldet planes: same keypoints and descriptors,
    for a single frame and for a batch, and the next extraction (hint grown) needs no retry."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        frames = np.stack([amd.synth_frame(640, 360, 70 + i) for i in range(3)])
        refs = [ref.extract(f) for f in frames]
        assert all(r.num_keypoints > 20 for r in refs)  # more keypoints (hence candidates) than the 16 list entries
        c.set_candidate_hint(16)
        assert_same_result(c.extract_features(frames[0]), refs[0], planes=False)
        c.set_candidate_hint(16)
        res = c.extract_features(torch.from_numpy(frames).cuda(), keep_all_planes=False)
        for i in range(3):
            assert_same_result(res, refs[i], planes=False, img=i)
        res = c.extract_features(torch.from_numpy(frames).cuda())
        assert_same_result(res, refs[1], planes=True, img=1)
    finally:
        c.close()


def test_all_pairs_match_over_rccl_world1(ctx, amd, ref):
    """The cross-GPU all-pairs path of BASELINE configs[4] with the real backend (RCCL, one rank on this box):
    device-resident 64-byte descriptor rows are gathered with all_gather and matched on the GPU; every pair equals
    the oracle's descriptor_match."""
    import torch
    import torch.distributed as dist
    frames = np.stack([amd.synth_frame(640, 360, 60 + i, shift=(4 * i, 2 * i)) for i in range(3)])
    res = ctx.extract_features(torch.from_numpy(frames).cuda(), keep_all_planes=False)
    refs = [ref.extract(f) for f in frames]
    local = []
    for i in range(3):
        total = res.counts(i)[1]
        rows = torch.zeros((total, 64), dtype=torch.uint8, device="cuda")
        rows[:, :61] = torch.from_numpy(res.descriptors(i)).cuda()
        local.append(rows)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    def match(a, b):  # device rows in, host records out
        out, cnt = ctx.descriptor_match_device(a, b, 10000, 0.86)
        ctx.synchronize()
        return out.cpu().numpy()[:int(cnt.item())].view(amd.MATCH_DTYPE).reshape(-1)

    def match_sets(q, cat, rows):  # one both-direction launch per lead image against the sets it leads
        out, cnt, cout, ccnt = ctx.descriptor_match_sets_mutual_device(q, cat, rows, 10000, 0.86)
        ctx.synchronize()
        out, cnt, cout, ccnt = out.cpu().numpy(), cnt.cpu().numpy(), cout.cpu().numpy(), ccnt.cpu().numpy()
        offs = np.concatenate([[0], np.cumsum(rows)])
        return ([out[k][:int(cnt[k])].copy().view(amd.MATCH_DTYPE).reshape(-1) for k in range(len(rows))],
                [cout[offs[k]:offs[k] + int(ccnt[k])].copy().view(amd.MATCH_DTYPE).reshape(-1) for k in range(len(rows))])

    try:
        pairs = amd.all_pairs_match(local, match)
        pairs_multi = amd.all_pairs_match(local, None, match_sets_fn=match_sets)
        torch.cuda.synchronize()
    finally:
        if created:
            dist.destroy_process_group()
    assert set(pairs) == {(i, j) for i in range(3) for j in range(3) if i != j} == set(pairs_multi)
    total = 0
    for (i, j), got in pairs.items():
        exp = ref.descriptor_match(refs[i].descriptors(), refs[j].descriptors(), 10000, 0.86)
        assert np.array_equal(got, exp), (i, j)
        assert np.array_equal(pairs_multi[(i, j)], exp), (i, j)
        total += len(exp)
    assert total > 20


def test_selection_on_the_device(amd, ref):
    """The order-dependent keypoint selection as dependency rounds on the device (k_select, akz_debug_set_select(2)): keypoints,
    extrema counts and descriptors identical to the host's grid selection on single frames, batches (the per-image runs are
    packed into one list), a 4K frame, a 5 x 5 pyramid, noise frames with dense candidates; a noise frame with more candidates
    than the kernel holds sends its job back to the host and still gives the same answer; one frame of each group against the
    oracle."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        noise = np.random.default_rng(11).integers(0, 256, (1, 270, 480), dtype=np.uint8)
        cases = [
            ("320x240", amd.synth_frame(320, 240, 5)[None], None, True, 2),
            ("1080p", amd.synth_frame(1920, 1080, 6)[None], None, True, 2),
            ("batch", np.stack([amd.synth_frame(640, 360, 70 + i) for i in range(5)]), None, True, 2),
            ("batch17", np.stack([amd.synth_frame(480, 270, 170 + i) for i in range(17)]), None, False, 2),
            ("4k", amd.synth_frame(3840, 2160, 2)[None], None, False, 2),
            ("5x5", amd.synth_frame(800, 600, 9)[None], dict(num_sublevels=5, max_octave_evolution=5), True, 2),
            ("flat", np.full((1, 240, 320), 90, np.uint8), None, False, 2),
            ("dense", noise, dict(detector_threshold=1e-7), True, 2),  # (before "noise": a short list would shorten the next job's)
            ("noise", noise, None, True, 2),
            ("many", np.random.default_rng(12).integers(0, 256, (1, 1200, 2048), dtype=np.uint8), None, False, None),
            ("batch+many", np.concatenate([amd.synth_frame(2048, 1200, 3)[None],
                                           np.random.default_rng(13).integers(0, 256, (1, 1200, 2048), dtype=np.uint8)]), None, False, None),
        ]
        fell_back = 0
        for name, frames, kw, check_oracle, where in cases:
            cfg = amd.Config(**kw) if kw else amd.Config()
            dev = torch.from_numpy(frames).cuda()
            c.debug_set_select(2)
            r2 = c.extract_features(dev, cfg, keep_all_planes=False)
            info = c.debug_select_info()
            if where is None:  # (more candidates than a workgroup's LDS holds: the job goes back to the host's selection)
                fell_back += info[0] != 2 and (info[2] & 0xffff) >= 1
            else:
                assert info[0] == where, (name, info)
                assert info[2] == 0 and (info[1] >= 1 or r2.counts(0)[1] == 0), (name, info)
            c.debug_set_select(False)
            r0 = c.extract_features(dev, cfg, keep_all_planes=False)
            assert c.debug_select_info()[0] == 0
            for i in range(len(frames)):
                assert r2.counts(i) == r0.counts(i), (name, i, r2.counts(i), r0.counts(i), info)
                assert r2.keypoints(i).tobytes() == r0.keypoints(i).tobytes(), (name, i)
                assert r2.descriptors(i).tobytes() == r0.descriptors(i).tobytes(), (name, i)
            if name != "flat":
                assert r2.counts(0)[1] > 10, name
            if check_oracle:
                i = len(frames) - 1
                q = ref.extract(frames[i], ref.default_config(**kw) if kw else None)
                assert r2.keypoints(i).tobytes() == q.keypoints().tobytes(), name
                assert np.array_equal(r2.descriptors(i), q.descriptors()), name
        assert fell_back >= 1
        # pipelined jobs, and a job with more keypoints than the one before it (the speculative fetch is too short)
        c.debug_set_select(2)
        small = torch.from_numpy(amd.synth_frame(320, 240, 5)[None]).cuda()
        one = torch.from_numpy(amd.synth_frame(1920, 1080, 6)[None]).cuda()
        c.debug_set_select(False)
        base = c.extract_features(one, keep_all_planes=False)
        c.debug_set_select(2)
        c.extract_features(small, keep_all_planes=False).close()
        jobs = [c.extract_begin(one, keep_all_planes=False) for _ in range(3)]
        for j in jobs:
            r = j.finish()
            assert r.keypoints(0).tobytes() == base.keypoints(0).tobytes() and r.descriptors(0).tobytes() == base.descriptors(0).tobytes()
        assert c.debug_select_info()[0] == 2
    finally:
        c.close()


def test_device_selection_is_not_retried_at_once_after_a_fallback(amd):
    """A job whose image has more candidates than k_select holds goes back to the host after the device's attempt; the next
    eight jobs of that shape leave the attempt out (jobs of another shape do not), the ninth tries again."""
    import torch
    many = torch.from_numpy(np.random.default_rng(12).integers(0, 256, (1, 1200, 2048), dtype=np.uint8)).cuda()
    other = torch.from_numpy(amd.synth_frame(640, 480, 1)[None]).cuda()
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        first = c.extract_features(many, keep_all_planes=False)  # (its list outgrows the first job's capacity: redone with room)
        want = (first.counts(0), first.keypoints(0).tobytes())
        assert c.debug_select_info()[3] > 65533
        attempts = []
        for k in range(11):
            c.extract_features(other, keep_all_planes=False).close()
            assert c.debug_select_info()[0] == 2  # (a synchronous call of another shape: on the device)
            r = c.extract_features(many, keep_all_planes=False)
            assert (r.counts(0), r.keypoints(0).tobytes()) == want
            r.close()
            info = c.debug_select_info()
            assert info[0] != 2
            attempts.append((info[2] & 0xffff) == 1)
        assert attempts == [True] + [False] * 8 + [True, False], attempts
    finally:
        c.close()


def test_selection_on_the_device_random_jobs(amd):
    """Randomised jobs (sizes, batch sizes, synthetic / noise / blended frames, thresholds, pyramid depths) through the device's
    selection and through the host's grids: identical counts, keypoints and descriptors, image by image."""
    import torch
    rng = np.random.default_rng(20261005)
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    on_device = 0
    try:
        for case in range(36):
            w, h = int(rng.integers(96, 720)), int(rng.integers(96, 560))
            n = int(rng.integers(1, 7))
            frames = []
            for i in range(n):
                kind = int(rng.integers(0, 4))
                syn = amd.synth_frame(w, h, int(rng.integers(0, 1000)))
                noi = rng.integers(0, 256, (h, w), dtype=np.uint8)
                frames.append(syn if kind == 0 else noi if kind == 1 else ((syn.astype(np.uint16) + noi) // 2).astype(np.uint8) if kind == 2
                              else np.full((h, w), int(rng.integers(0, 256)), np.uint8))
            frames = np.stack(frames)
            kw = {}
            if rng.integers(0, 3) == 0:
                kw["detector_threshold"] = float(10.0 ** rng.uniform(-6, -2.5))
            if rng.integers(0, 4) == 0 and min(w, h) >= 200:
                kw.update(num_sublevels=int(rng.integers(2, 6)), max_octave_evolution=int(rng.integers(2, 5)))
            cfg = amd.Config(**kw)
            dev = torch.from_numpy(frames).cuda()
            c.debug_set_select(2)
            r2 = c.extract_features(dev, cfg, keep_all_planes=False)
            on_device += c.debug_select_info()[0] == 2
            c.debug_set_select(0)
            r0 = c.extract_features(dev, cfg, keep_all_planes=False)
            for i in range(n):
                assert r2.counts(i) == r0.counts(i), (case, w, h, n, kw, i, r2.counts(i), r0.counts(i))
                assert r2.keypoints(i).tobytes() == r0.keypoints(i).tobytes(), (case, w, h, n, kw, i)
                assert r2.descriptors(i).tobytes() == r0.descriptors(i).tobytes(), (case, w, h, n, kw, i)
            r2.close(); r0.close()
        assert on_device >= 30, on_device
    finally:
        c.close()


def test_selection_on_the_device_from_four_threads(amd):
    """Four host threads, a context and a stream each, synchronous calls with the device's selection at the same time (the
    workgroups of k_select / k_sort_rows of different contexts compete for compute units with most of their LDS free): every
    result identical to the one a lone context gives."""
    import threading
    import torch
    frames = [amd.synth_frame(960, 540, 300 + i)[None] for i in range(4)] + [amd.synth_frame(1920, 1080, 310)[None]]
    c0 = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        c0.debug_set_select(0)
        expect = []
        for fr in frames:
            r = c0.extract_features(torch.from_numpy(fr).cuda(), keep_all_planes=False)
            expect.append((r.counts(0), r.keypoints(0).tobytes(), r.descriptors(0).tobytes()))
            r.close()
    finally:
        c0.close()
    dev_frames = [torch.from_numpy(fr).cuda() for fr in frames]
    torch.cuda.synchronize()
    errors, bar = [], threading.Barrier(4)

    def worker(k):
        try:
            st = torch.cuda.Stream()
            c = amd.Context(0, st.cuda_stream)
            try:
                c.debug_set_select(2)
                bar.wait()
                for it in range(12):
                    j = (k + it) % len(frames)
                    r = c.extract_features(dev_frames[j], keep_all_planes=False)
                    if (r.counts(0), r.keypoints(0).tobytes(), r.descriptors(0).tobytes()) != expect[j]:
                        errors.append((k, it, j, r.counts(0), expect[j][0]))
                    if c.debug_select_info()[0] != 2:
                        errors.append((k, it, j, "not on the device", c.debug_select_info()))
                    r.close()
            finally:
                c.close()
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads)
    assert not errors, errors[:4]


def test_short_candidate_lists_take_the_one_launch_sort_and_overflow_back(amd):
    """A job of the same shape as the one before it whose list was short gets a list no longer than the one-launch sort takes
    (k_sort_small); a frame of that shape with more candidates than that overflows it and is redone with room: every result
    identical to a fresh context's."""
    import torch
    quiet = [amd.synth_frame(960, 540, 40 + i)[None] for i in range(3)]
    busy = np.random.default_rng(5).integers(0, 256, (1, 540, 960), dtype=np.uint8)
    seq = [quiet[0], quiet[1], busy, quiet[2], quiet[0], busy, busy, quiet[1]]
    expect = []
    for fr in seq:
        c0 = amd.Context(0, torch.cuda.current_stream().cuda_stream)
        try:
            r = c0.extract_features(torch.from_numpy(fr).cuda(), keep_all_planes=False)
            expect.append((r.counts(0), r.keypoints(0).tobytes(), r.descriptors(0).tobytes()))
        finally:
            c0.close()
    assert expect[2][0][1] > 3000 > expect[0][0][1] > 50
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        for mode in (None, 2, 1):
            c.debug_set_select(mode)
            for k, fr in enumerate(seq):
                r = c.extract_features(torch.from_numpy(fr).cuda(), keep_all_planes=False)
                assert (r.counts(0), r.keypoints(0).tobytes(), r.descriptors(0).tobytes()) == expect[k], (mode, k, r.counts(0), expect[k][0])
            jobs = [c.extract_begin(torch.from_numpy(fr).cuda(), keep_all_planes=False) for fr in (quiet[0], quiet[1], busy)]
            for k, j in zip((0, 1, 2), jobs):
                r = j.finish()
                assert (r.counts(0), r.keypoints(0).tobytes(), r.descriptors(0).tobytes()) == expect[k], (mode, "pipelined", k)
    finally:
        c.close()


def test_selection_from_device_neighbour_lists(amd, ref):
    """The host's order-dependent keypoint selection fed by the device's neighbour lists (k_relations: who can be within
    `size` of whom; akz_debug_set_select(1)) against the spatial-grid form (0): identical keypoints and descriptors on
    single frames, a batch, a 4K frame, a 5 x 5 pyramid and a noise frame whose dense candidates overflow the lists (those
    images fall back to the grids); one frame of each group against the oracle."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    try:
        cases = [
            ("320x240", amd.synth_frame(320, 240, 5)[None], None, True),
            ("1080p", amd.synth_frame(1920, 1080, 6)[None], None, True),
            ("batch", np.stack([amd.synth_frame(640, 360, 70 + i) for i in range(5)]), None, True),
            ("4k", amd.synth_frame(3840, 2160, 2)[None], None, False),
            ("5x5", amd.synth_frame(800, 600, 9)[None], dict(num_sublevels=5, max_octave_evolution=5), True),
            ("noise", np.random.default_rng(11).integers(0, 256, (1, 270, 480), dtype=np.uint8), None, True),
        ]
        for name, frames, kw, check_oracle in cases:
            cfg = amd.Config(**kw) if kw else amd.Config()
            dev = torch.from_numpy(frames).cuda()
            c.debug_set_select(True)
            r1 = c.extract_features(dev, cfg, keep_all_planes=False)
            c.debug_set_select(False)
            r0 = c.extract_features(dev, cfg, keep_all_planes=False)
            for i in range(len(frames)):
                assert r1.counts(i) == r0.counts(i), (name, i, r1.counts(i), r0.counts(i))
                assert r1.keypoints(i).tobytes() == r0.keypoints(i).tobytes(), (name, i)
                assert r1.descriptors(i).tobytes() == r0.descriptors(i).tobytes(), (name, i)
            assert r1.counts(0)[1] > 10, name
            if check_oracle:
                i = len(frames) - 1
                q = ref.extract(frames[i], ref.default_config(**kw) if kw else None)
                assert r1.keypoints(i).tobytes() == q.keypoints().tobytes(), name
                assert np.array_equal(r1.descriptors(i), q.descriptors()), name
        # pipelined jobs through the lists (lone frames take them by default)
        c.debug_set_select(None)
        one = torch.from_numpy(amd.synth_frame(1920, 1080, 6)[None]).cuda()
        base = c.extract_features(one, keep_all_planes=False)
        jobs = [c.extract_begin(one, keep_all_planes=False) for _ in range(3)]
        for j in jobs:
            r = j.finish()
            assert r.keypoints(0).tobytes() == base.keypoints(0).tobytes() and r.descriptors(0).tobytes() == base.descriptors(0).tobytes()
    finally:
        c.close()


def test_describe_caller_supplied_keypoints(ctx, amd, ref):
    """compute_main_orientation + extract_descriptors as stand-alone ops (scale_space_extrema.rs:207-329,
    descriptors.rs:14-35) on keypoints the detector did not produce: the detector's own list reordered, and moved /
    rescaled copies of it, against the oracle on its pyramid."""
    frame = amd.synth_frame(800, 600, 21)
    res, rf = ctx.extract_features(frame), ref.extract(frame)
    kp = res.keypoints()
    assert len(kp) > 100
    # 1) the detector's own keypoints, reversed, with the stored angle: same descriptors row for row
    k1, d1 = res.describe_keypoints(kp[::-1], compute_orientation=False)
    assert np.array_equal(d1, res.descriptors()[::-1]) and k1.tobytes() == kp[::-1].tobytes()
    # 2) ... and with the orientation recomputed: same angles
    k2, d2 = res.describe_keypoints(kp[::-1], compute_orientation=True)
    assert np.array_equal(k2["angle"], kp["angle"][::-1]) and np.array_equal(d2, d1)
    # 3) moved and rescaled keypoints (interior ones, so that every sample stays inside the image)
    inner = kp[(kp["x"] > 150) & (kp["x"] < 650) & (kp["y"] > 150) & (kp["y"] < 450)][:200].copy()
    inner["x"] += np.float32(1.3)
    inner["y"] -= np.float32(0.7)
    inner["size"] *= np.float32(1.1)
    inner["angle"] = np.float32(0.37)
    for orient in (True, False):
        got_k, got_d = res.describe_keypoints(inner, compute_orientation=orient)
        exp_k, exp_d = rf.describe(inner, compute_orientation=orient)
        assert np.array_equal(got_k["angle"], exp_k["angle"]), orient
        assert np.array_equal(got_d, exp_d), orient
    assert not np.array_equal(got_k["angle"], res.describe_keypoints(inner, compute_orientation=True)[0]["angle"])
    bad = inner[:2].copy()
    bad["class_id"] = 99
    with pytest.raises(amd.AkazeError):
        res.describe_keypoints(bad)


def test_two_contexts_on_two_host_threads(amd, ref):
    """One context per host thread (the C ABI's threading rule), two threads on the same GPU at once: results are
    those of the oracle, nothing leaks between the contexts."""
    import threading
    import torch
    frames = [np.stack([amd.synth_frame(640, 360, 80 + 10 * t + i) for i in range(2)]) for t in range(2)]
    out, errs = [None, None], []

    def work(t):
        try:
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                c = amd.Context(0, stream.cuda_stream)
                d = torch.from_numpy(frames[t]).cuda()
                stream.synchronize()
                last = None
                for _ in range(6):
                    last = c.extract_features(d, keep_all_planes=False)
                out[t] = [(last.keypoints(i).copy(), last.descriptors(i).copy()) for i in range(2)]
                c.close()
        except Exception as e:  # surfaced in the main thread
            errs.append(e)

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    [x.start() for x in th]
    [x.join() for x in th]
    assert not errs, errs
    for t in range(2):
        for i in range(2):
            q = ref.extract(frames[t][i])
            assert out[t][i][0].tobytes() == q.keypoints().tobytes() and np.array_equal(out[t][i][1], q.descriptors())


def test_result_may_outlive_its_context(amd, ref):
    """Freeing in the 'wrong' order is safe: a result keeps its host data after akz_ctx_destroy, device accessors
    report the destroyed context instead of touching freed memory, and the result can still be freed."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    frame = amd.synth_frame(480, 270, 3)
    res = c.extract_features(frame)
    c.set_eager_finish(False)
    job = c.extract_begin(torch.from_numpy(frame[None]).cuda())      # its finish half would run in finish(): refused later
    c.set_eager_finish(True)
    job_e = c.extract_begin(torch.from_numpy(frame[None]).cuda())    # finished by the context's own thread before it goes
    c.close()
    q = ref.extract(frame)
    assert res.keypoints().tobytes() == q.keypoints().tobytes() and np.array_equal(res.descriptors(), q.descriptors())
    with pytest.raises(amd.AkazeError):
        res.plane(2, "Lt")
    with pytest.raises(amd.AkazeError):
        job.finish()
    late = job_e.finish()
    assert late.keypoints().tobytes() == q.keypoints().tobytes() and np.array_equal(late.descriptors(), q.descriptors())
    with pytest.raises(amd.AkazeError):
        late.plane(2, "Lt")
    late.close()
    res.close()
    del job


def test_ops_on_caller_held_evolutions(ctx, amd, ref):
    """`pub mod ops` on evolutions the caller holds (ops::scale_space_extrema::detect_keypoints,
    ops::descriptors::extract_descriptors): the ORACLE's planes, uploaded through akz_extract_from_planes, give the
    oracle's keypoints (with orientation) and descriptors; with detection switched off the pyramid still serves
    caller-supplied keypoints."""
    frame = amd.synth_frame(517, 389, 31)
    rf = ref.extract(frame)
    planes = [{pl: rf.plane(lvl, pl) for pl in PLANES} for lvl in range(rf.num_levels)]
    res = ctx.extract_from_planes(517, 389, planes)
    kp, rk = res.keypoints(), rf.keypoints()
    assert len(kp) == rf.num_keypoints > 50
    for f in KP_FIELDS:
        assert np.array_equal(kp[f], rk[f]), f
    assert np.array_equal(res.descriptors(), rf.descriptors())
    for lvl in (0, 5, rf.num_levels - 1):
        assert np.array_equal(res.plane(lvl, "Ldet"), rf.plane(lvl, "Ldet"))
    # only what the two ops read (detect_keypoints: Ldet, Lx, Ly; extract_descriptors: Lt, Lx, Ly)
    slim = [{pl: rf.plane(lvl, pl) for pl in ("Lt", "Lx", "Ly")} for lvl in range(rf.num_levels)]
    r2 = ctx.extract_from_planes(517, 389, slim, detect=False)
    assert r2.counts()[1] == 0
    k2, d2 = r2.describe_keypoints(rk, compute_orientation=False)
    assert np.array_equal(d2, rf.descriptors())
    with pytest.raises(amd.AkazeError):
        ctx.extract_from_planes(517, 389, slim)          # detection needs Ldet
    with pytest.raises(amd.AkazeError):
        ctx.extract_from_planes(517, 389, planes[:-1])   # wrong number of evolutions
    res.close()
    r2.close()


def test_host_frames_begin_matches_device_frames(amd, ref):
    """akz_extract_begin_host_*: frames in (pinned) host memory, uploaded on the context's copy stream into per-slot
    staging buffers.  Three jobs in flight with three DIFFERENT batches (a staging buffer that was reused too early
    would mix them up), then a second round that reuses the slots; u8 and f32; byte-identical to device input."""
    import torch
    c = amd.Context(0, torch.cuda.Stream().cuda_stream)
    try:
        sets = [np.stack([amd.synth_frame(400, 300, 10 * k + i) for i in range(2)]) for k in range(3)]
        exp = [c.extract_features(torch.from_numpy(s).cuda(), keep_all_planes=False) for s in sets]
        for rnd in range(2):
            hosts = [torch.from_numpy(s).pin_memory() if rnd == 0 else s.copy() for s in sets]  # pinned, then pageable
            jobs = [c.extract_begin_host(h, keep_all_planes=False) for h in hosts]
            got = [j.finish() for j in jobs]
            for g, e in zip(got, exp):
                for i in range(2):
                    assert g.keypoints(i).tobytes() == e.keypoints(i).tobytes()
                    assert g.descriptors(i).tobytes() == e.descriptors(i).tobytes()
                    assert g.contrast(i) == e.contrast(i)
        f32 = (sets[1].astype(np.float32) * np.float32(1.0)) / np.float32(255.0)
        g = c.extract_begin_host(f32).finish()
        assert_same_result(g, ref.extract(f32[1]), img=1)
        c.set_eager_finish(False)  # (jobs finished by the context's own thread give their slots back on their own)
        with pytest.raises(amd.AkazeError):
            [c.extract_begin_host(sets[0]) for _ in range(4)]  # a fourth job in flight is refused
    finally:
        c.close()


@pytest.mark.parametrize("host_sort", [True, False])
def test_candidate_order_host_and_device_sort(amd, ref, host_sort):
    """The unordered candidate list is put into the reference's scan order either on the host (bucket by image, counting
    sort) or on the device (radix sort by image, level, pixel): a batch with different candidate counts per image, an
    image without any candidate, and a second call that reuses the buffers."""
    import torch
    c = amd.Context(0, torch.cuda.Stream().cuda_stream)
    c.debug_set_host_sort(host_sort)
    try:
        frames = np.stack([amd.synth_frame(486, 270, 30 + i) for i in range(4)])
        frames[2] = 128  # flat: no candidates at all
        for _ in range(2):
            res = c.extract_features(torch.from_numpy(frames).cuda(), keep_all_planes=False)
            for i in range(4):
                assert_same_result(res, ref.extract(frames[i]), planes=False, img=i, equal_nan=True)
            res.close()
        one = amd.synth_frame(1001, 300, 4)
        assert_same_result(c.extract_features(one, keep_all_planes=False), ref.extract(one), planes=False)
    finally:
        c.close()
