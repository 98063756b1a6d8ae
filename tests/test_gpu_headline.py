"""Oracle parity at the EXACT geometries bench.py times (round-4 verdict, item 3).  Band heights, strip counts, the fork of the
coarse chain and the resident tail all depend on the batch size, so the small-batch tests elsewhere do not cover the launches
of the headline step: here the bench's own batches go through the bench's own calls (extract_begin with AKZ_INPUT_READY on
resident frames, two batches in flight; extract_begin_host for the pair) and sampled frames are compared with the CPU oracle
(reference entry points: akaze/src/lib.rs:167-194 extract_features, :252-275 match_features).  Bar: identical."""
import numpy as np
import pytest

from test_gpu_extract import PLANES, assert_same_result

pytestmark = pytest.mark.gpu


def _planes_equal(res, rf, img, picks):
    for lvl, pl in picks:
        a, b = res.plane(lvl, pl, img), rf.plane(lvl, pl)
        assert a.shape == b.shape and np.array_equal(a, b), (img, lvl, pl)


def test_headline_32x1080p_device_batch_all_planes(ctx, amd, ref):
    """bench.py's default step: 32 x 1920x1080 frames (synthetic frames 0..31) resident in HBM, Config::default(), all ten
    EvolutionStep planes kept, batch j+1 begun before batch j is finished.  Frames 0, 15 and 31 of the FIRST batch: every
    plane of every level, keypoints, descriptors; frame 7 of the second batch: keypoints and descriptors."""
    import torch
    frames = np.stack([amd.synth_frame(1920, 1080, i) for i in range(32)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    cfg = amd.Config()
    j1 = ctx.extract_begin(d, cfg, keep_all_planes=True, input_ready=True)
    j2 = ctx.extract_begin(d, cfg, keep_all_planes=True, input_ready=True)
    r1 = j1.finish()
    r2 = j2.finish()
    assert r1.num_images == 32
    for img in (0, 15, 31):
        rf = ref.extract(frames[img], threads=8)
        assert rf.num_levels == 16 and rf.num_keypoints > 1000
        assert_same_result(r1, rf, planes=True, img=img)
        rf.close()
    rf = ref.extract(frames[7], threads=8)
    assert_same_result(r2, rf, planes=False, img=7)
    for img in range(32):  # the two batches agree frame by frame
        assert r1.keypoints(img).tobytes() == r2.keypoints(img).tobytes()
        assert np.array_equal(r1.descriptors(img), r2.descriptors(img))
    r1.close()
    r2.close()


@pytest.mark.parametrize("sub,octv", [(4, 4), (5, 5)])
def test_headline_8x4k_device_batch(ctx, amd, ref, sub, octv):
    """The 8 x 3840x2160 step (DESIGN's 4K rows; 5 x 5 is BASELINE configs[4]'s extraction): frames 0 and 7 -- keypoints,
    descriptors and a plane of each kind from the first, a middle and the last level."""
    import torch
    frames = np.stack([amd.synth_frame(3840, 2160, i) for i in range(8)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    cfg = amd.Config(num_sublevels=sub, max_octave_evolution=octv)
    res = ctx.extract_begin(d, cfg, keep_all_planes=True, input_ready=True).finish()
    rcfg = ref.default_config(num_sublevels=sub, max_octave_evolution=octv)
    for img in (0, 7):
        rf = ref.extract(frames[img], rcfg, threads=8)
        L = rf.num_levels
        assert L == sub * octv and rf.num_keypoints > 4000
        assert_same_result(res, rf, planes=False, img=img)
        _planes_equal(res, rf, img, [(0, "Lt"), (1, "Lsmooth"), (sub - 1, "Lflow"), (sub, "Lt"), (2 * sub - 1, "Ldet"),
                                     (2 * sub, "Lstep"), (L - 1, "Lt"), (L - 1, "Ldet"), (L - 2, "Lxy"), (3, "Lx")])
        rf.close()
    res.close()


def test_pair_4k_host_batch_then_match_features(ctx, amd, ref):
    """bench.py's pair_4k leg (BASELINE configs[2]): ONE extract call on the two-frame batch in pinned HOST memory -- which takes
    the march kernels, unlike two single-frame calls -- then match_features with its reference signature."""
    import torch
    pair = np.stack([amd.synth_frame(3840, 2160, 0), amd.synth_frame(3840, 2160, 0, shift=(17, 9))])
    h_pair = torch.from_numpy(pair).pin_memory()
    rp = ctx.extract_begin_host(h_pair, amd.Config(), keep_all_planes=True).finish()
    q = [ref.extract(pair[i], threads=8) for i in range(2)]
    for i in range(2):
        assert_same_result(rp, q[i], planes=False, img=i)
        _planes_equal(rp, q[i], i, [(0, "Lt"), (2, "Lflow"), (3, "Lxx"), (5, "Lsmooth"), (7, "Ldet"), (11, "Lstep"), (15, "Lt"), (15, "Ldet")])
    k0, k1, d0, d1 = rp.keypoints(0), rp.keypoints(1), rp.descriptors(0), rp.descriptors(1)
    raw = ref.descriptor_match(q[0].descriptors(), q[1].descriptors(), 10000, 0.86)
    assert len(raw) > 500 and np.array_equal(ctx.descriptor_match(d0, d1, 10000, 0.86), raw)
    amd.random_seed(42, 69)
    ref.random_seed(42, 69)
    got = amd.match_features(k0, d0, k1, d1, 0.86, 1000, 3.0, ctx=ctx)
    exp = ref.remove_outliers(q[0].keypoints(), q[1].keypoints(), raw, 1000, 0.05, 3.0)
    assert np.array_equal(got, exp) and 8 <= len(got) <= len(raw)
    rp.close()


@pytest.mark.parametrize("w,h,n", [(2016, 1512, 1), (1920, 1080, 2), (1920, 1080, 3), (3840, 2160, 1), (1600, 1200, 1), (1366, 768, 1),
                                   (1920, 1080, 6)])
def test_job_size_gates_both_entry_points(ctx, amd, ref, w, h, n):
    """Jobs either side of the job gates (csrc/akz_gates.hpp: big_px_sync / big_px_async, 1.4 Mpx since the end of round 6;
    tiled_prep_px 11 Mpx) through BOTH entry points: the batch path -- forked coarse chain, marches and resident tail where the
    size allows -- from 1.4 Mpx, the tiled preparation family below 11 Mpx (a lone 4K frame), the march family above (6 x 1080p);
    1366 x 768 (1.05 Mpx) is a one-stream chain as a call and a batch-path job when begun (big_px_async: 0.3 Mpx).  Both must equal the oracle, planes included (lib.rs:167-194)."""
    import torch
    frames = np.stack([amd.synth_frame(w, h, 40 + i) for i in range(n)])
    d = torch.from_numpy(frames).cuda()
    torch.cuda.synchronize()
    sync = ctx.extract_features(d, keep_all_planes=True)
    asyn = ctx.extract_begin(d, amd.Config(), keep_all_planes=True).finish()
    for img in range(n):
        rf = ref.extract(frames[img], threads=8)
        L = rf.num_levels
        for res in (sync, asyn):
            assert_same_result(res, rf, planes=False, img=img)
            _planes_equal(res, rf, img, [(0, "Lt"), (1, "Lflow"), (2, "Lsmooth"), (3, "Ldet"), (3, "Lxx"), (4, "Lt"), (5, "Lstep"),
                                         (7, "Ly"), (8, "Lt"), (L - 1, "Lt"), (L - 1, "Ldet"), (L - 2, "Lflow")])
        rf.close()
    sync.close()
    asyn.close()
