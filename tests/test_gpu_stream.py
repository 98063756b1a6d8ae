"""The streaming (register-ring) kernels of akz_stream.hip and the detector column march of akz_march.hip against the
CPU oracle and against the LDS-tiled kernels: every plane, keypoint and descriptor byte identical, for shapes that
exercise the strip / band / edge-column logic (widths that are not multiples of 4 or of the strips, several strips
and bands, batches)."""
import os

import numpy as np
import pytest

from test_gpu_extract import PLANES, assert_same_result

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[(4, 1), (5, 3)], ids=["tiled_fused+stream_prep", "march+level_march"])
def sctx(amd, request):
    """A context with the streaming preparation / contrast / blur kernels forced on and one of the one-kernel detector
    forms forced (the LDS-tiled kernel or the column march), so that small test images take them too."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    c.set_detector_mode(request.param[0])
    c.set_prep_mode(request.param[1])  # 1: streaming preparation kernel; 3: preparation fused with the first diffusion steps
    yield c
    c.close()


@pytest.mark.parametrize("w,h,idx", [(320, 240, 0), (517, 389, 2), (249, 131, 3), (1001, 300, 4)])
def test_stream_extract_matches_oracle_all_planes(sctx, amd, ref, w, h, idx):
    frame = amd.synth_frame(w, h, idx)
    assert_same_result(sctx.extract_features(frame), ref.extract(frame))


def test_random_shapes_all_planes(sctx, amd, ref):
    """A seeded sweep over image shapes (widths around the 480-column strips and the 8 x 8 patches, heights that put band
    seams at arbitrary rows, odd sizes, flat and tall images): every plane, keypoint and descriptor byte against the
    oracle through the forced kernels (march / level march / contrast march / resident, or tiled + streaming)."""
    rng = np.random.default_rng(20261004)
    shapes = [(int(rng.integers(96, 1100)), int(rng.integers(64, 700))) for _ in range(8)]
    shapes += [(479, 97), (481, 133), (961, 66), (1441, 81), (130, 641)]
    for k, (w, h) in enumerate(shapes):
        frame = amd.synth_frame(w, h, 100 + k)
        assert_same_result(sctx.extract_features(frame), ref.extract(frame, threads=8)), (w, h)


def test_stream_batch_and_nondefault_config(sctx, amd, ref):
    import torch
    frames = np.stack([amd.synth_frame(486, 270, i) for i in range(3)])
    res = sctx.extract_features(torch.from_numpy(frames).cuda())
    for i in range(3):
        assert_same_result(res, ref.extract(frames[i]), planes=(i == 2), img=i)
    frame = amd.synth_frame(640, 360, 7)
    for kw in (dict(num_sublevels=5, max_octave_evolution=5), dict(detector_threshold=0.0005, num_sublevels=3)):
        assert_same_result(sctx.extract_features(frame, amd.Config(**kw)), ref.extract(frame, ref.default_config(**kw)),
                           planes=False)


@pytest.mark.parametrize("sigma", [1, 2, 3, 4])
def test_stream_detector_response_op(sctx, ref, sigma):
    import torch
    rng = np.random.default_rng(40 + sigma)
    for (h, w) in ((96, 130), (70, 517)):
        ls = rng.random((h, w), dtype=np.float32)
        got = sctx.detector_response(torch.from_numpy(ls).cuda(), sigma)
        lx = ref.scharr(ls, True, False, sigma)
        ly = ref.scharr(ls, False, True, sigma)
        exp = {"Lx": lx, "Ly": ly, "Lxx": ref.scharr(lx, True, False, sigma), "Lyy": ref.scharr(ly, False, True, sigma),
               "Lxy": ref.scharr(lx, False, True, sigma)}
        exp["Ldet"] = ((exp["Lxx"] * exp["Lyy"]) - (exp["Lxy"] * exp["Lxy"])) * np.float32(sigma ** 4)
        for k, v in exp.items():
            assert np.array_equal(got[k].cpu().numpy().reshape(h, w), v), (sigma, w, h, k)
        lean = sctx.detector_response(torch.from_numpy(ls).cuda(), sigma, keep_second=False)
        assert np.array_equal(lean["Ldet"].cpu().numpy().reshape(h, w), exp["Ldet"])


@pytest.mark.parametrize("w,h", [(480, 72), (481, 40), (479, 33), (961, 50), (1441, 129), (24, 300), (41, 25)])
def test_march_strip_and_band_edges(amd, ref, w, h):
    """The column march on widths around its 480-column strips (one column into the next strip, one short of it, three
    strips + 1), the smallest supported sizes and tall narrow images (several row bands): all six planes, every
    sigma_size, against the oracle."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    c.set_detector_mode(5)
    try:
        rng = np.random.default_rng(w * 1000 + h)
        ls = rng.random((2, h, w), dtype=np.float32)
        for sigma in (1, 2, 3, 4):
            if w < 4 * sigma + 8 or h < 4 * sigma + 8:
                continue
            got = c.detector_response(torch.from_numpy(ls).cuda(), sigma)
            for i in range(2):
                lx = ref.scharr(ls[i], True, False, sigma)
                ly = ref.scharr(ls[i], False, True, sigma)
                exp = {"Lx": lx, "Ly": ly, "Lxx": ref.scharr(lx, True, False, sigma), "Lyy": ref.scharr(ly, False, True, sigma),
                       "Lxy": ref.scharr(lx, False, True, sigma)}
                exp["Ldet"] = ((exp["Lxx"] * exp["Lyy"]) - (exp["Lxy"] * exp["Lxy"])) * np.float32(sigma ** 4)
                for k, v in exp.items():
                    assert np.array_equal(got[k][i].cpu().numpy(), v), (sigma, w, h, i, k)
    finally:
        c.close()


@pytest.mark.parametrize("w,h,idx", [(962, 300, 11), (481, 270, 12), (1500, 200, 13)])
def test_march_extract_candidates_across_strips(amd, ref, w, h, idx):
    """Whole extraction with the march forced: extrema candidates next to strip and band boundaries must be reported
    exactly once (keypoints, descriptors and every plane identical to the oracle)."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    c.set_detector_mode(5)
    try:
        frame = amd.synth_frame(w, h, idx)
        assert_same_result(c.extract_features(frame), ref.extract(frame, threads=8))
    finally:
        c.close()


@pytest.mark.parametrize("w,h,idx", [(480, 64, 21), (481, 97, 22), (479, 130, 23), (961, 66, 24), (1445, 160, 25), (16, 300, 26),
                                     (35, 16, 27)])
def test_level_march_strip_and_band_edges(amd, ref, w, h, idx):
    """The fused level kernel (preparation + first diffusion steps) on widths around its 480-column strips, minimal
    sizes and tall narrow images (several row bands), default and 5 x 5 configurations (2 .. 4 fused steps, levels with
    more than four steps continue in k_fed_own, new octaves start from a materialised 2x2 mean), with and without
    Lstep: every plane of every level, keypoints and descriptors against the oracle."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    c.set_prep_mode(3)
    try:
        frame = amd.synth_frame(w, h, idx)
        assert_same_result(c.extract_features(frame), ref.extract(frame), equal_nan=True)
        kw = dict(num_sublevels=5, max_octave_evolution=5)
        assert_same_result(c.extract_features(frame, amd.Config(**kw)), ref.extract(frame, ref.default_config(**kw)), equal_nan=True)
        lean = c.extract_features(frame, keep_all_planes=False)
        assert_same_result(lean, ref.extract(frame), equal_nan=True)  # Lstep recomputed on fetch
    finally:
        c.close()


def test_stream_equals_tiled_on_1080p_batch(ctx, sctx, amd):
    """Full-size property check (the oracle would take minutes): both kernel families give the same bytes."""
    import torch
    frames = torch.from_numpy(np.stack([amd.synth_frame(1920, 1080, 20 + i) for i in range(4)])).cuda()
    ctx.set_detector_mode(0)
    ctx.set_prep_mode(0)
    try:
        a = ctx.extract_features(frames)
    finally:
        ctx.set_detector_mode(2)
        ctx.set_prep_mode(2)
    b = sctx.extract_features(frames)
    d = ctx.extract_features(frames)  # the default (automatic) choice
    for i in range(4):
        assert a.counts(i) == b.counts(i) == d.counts(i) and a.counts(i)[1] > 1000
        assert a.keypoints(i).tobytes() == b.keypoints(i).tobytes() == d.keypoints(i).tobytes()
        assert np.array_equal(a.descriptors(i), b.descriptors(i)) and np.array_equal(a.descriptors(i), d.descriptors(i))
    for lvl in (0, 1, 3, 4, 9, 15):
        for pl in PLANES:
            x = a.plane(lvl, pl, 1)
            assert np.array_equal(x, b.plane(lvl, pl, 1)), (lvl, pl)
            assert np.array_equal(x, d.plane(lvl, pl, 1)), (lvl, pl)


@pytest.mark.parametrize("shape", [(96, 130), (131, 249), (300, 517), (64, 1001)])
def test_stream_contrast_factor(sctx, ref, shape):
    """compute_contrast_factor through the streaming passes (no blurred plane): identical f64 result, for several
    bin counts / percentiles and a batch whose images have different maxima."""
    import torch
    rng = np.random.default_rng(shape[1])
    imgs = np.stack([ref.gaussian_blur(rng.random(shape, dtype=np.float32) * np.float32(s), 1.6) for s in (1.0, 0.3, 0.05)])
    for nbins, pct in ((300, 0.7), (64, 0.5), (1000, 0.9), (301, 0.7), (77, 0.35)):  # odd counts too: the f64 thresholds follow the histograms in LDS
        got = sctx.contrast_factor(torch.from_numpy(imgs).cuda(), pct, 1.0, nbins).cpu().numpy()
        for i in range(3):
            assert float(got[i]) == ref.contrast_factor(imgs[i], pct, 1.0, nbins), (nbins, pct, i)
    flat = np.full(shape, 0.25, np.float32)
    assert float(sctx.contrast_factor(torch.from_numpy(flat).cuda()).cpu().numpy()[0]) == ref.contrast_factor(flat)


@pytest.mark.parametrize("shape", [(700, 520), (200, 480), (333, 961), (129, 1440), (16, 16), (65, 481)])
def test_contrast_march(amd, ref, shape):
    """k_contrast_march (the two contrast passes as column marches; prep mode 3 takes them for any size): images cut into
    several bands (band seams inside the image), widths at and around the 480-column strips (one strip exactly, strips
    with a single owned column, three strips), odd widths (dword accesses), the smallest supported image; a batch whose
    images have different maxima, several bin counts and percentiles; a constant image (no gradient: 0.03)."""
    import torch
    c = amd.Context(0, torch.cuda.current_stream().cuda_stream)
    c.set_prep_mode(3)
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    imgs = np.stack([ref.gaussian_blur(rng.random(shape, dtype=np.float32) * np.float32(s), 1.6) for s in (1.0, 0.3, 0.05)])
    for nbins, pct in ((300, 0.7), (64, 0.5), (640, 0.9), (301, 0.7), (1, 0.7)):
        got = c.contrast_factor(torch.from_numpy(imgs).cuda(), pct, 1.0, nbins).cpu().numpy()
        for i in range(3):
            assert float(got[i]) == ref.contrast_factor(imgs[i], pct, 1.0, nbins), (nbins, pct, i)
    flat = np.full(shape, 0.25, np.float32)
    assert float(c.contrast_factor(torch.from_numpy(flat).cuda()).cpu().numpy()[0]) == ref.contrast_factor(flat)
    c.close()


@pytest.mark.parametrize("shape", [(96, 132), (131, 248), (77, 516), (40, 1000), (300, 517)])
def test_stream_blur5(sctx, ref, shape):
    """The level-0 blur (sigma 1.6 -> 5 taps) through the streaming kernel: f32 input for any width, u8 input (with the
    tabulated u8 -> unit float conversion) for 4-byte aligned rows; other cases fall back to the tiled kernel."""
    import torch
    rng = np.random.default_rng(shape[0])
    f = rng.random((2,) + shape, dtype=np.float32)
    got = sctx.gaussian_blur(torch.from_numpy(f).cuda(), 1.6).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], ref.gaussian_blur(f[i], 1.6)), i
    u8 = rng.integers(0, 256, (2,) + shape, dtype=np.uint8)
    unit = (u8.astype(np.float32) * np.float32(1.0)) / np.float32(255.0)  # image.rs:136
    got = sctx.gaussian_blur(torch.from_numpy(u8).cuda(), 1.6).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], ref.gaussian_blur(unit[i], 1.6)), i


def test_auto_mode_odd_width_batch_matches_oracle(ctx, amd, ref):
    """The automatic kernel choice on a batch that is large enough for the streaming kernels but has rows that are
    not multiples of 4 pixels (u8 level-0 blur falls back to the tiled kernel, streaming preparation and the
    grouped coarse-level detector run): every plane of one frame and all keypoints / descriptors."""
    import torch
    frames = np.stack([amd.synth_frame(1283, 721, 70 + i) for i in range(3)])
    res = ctx.extract_features(torch.from_numpy(frames).cuda())
    for i in range(3):
        assert_same_result(res, ref.extract(frames[i], threads=8), planes=(i == 1), img=i)
