"""Image ingest, options files and debug output: host code of libakaze_hip.so, no GPU.

The reference delegates decoding to the `image` crate; for lossy input parity with it is unpinned (see
akz_image.cpp).  Here the decoders are checked against an independent decoder (Pillow / libjpeg-turbo): lossless
formats must agree exactly, JPEG within the usual IDCT / upsampling rounding differences.  The two JPEG files
under tests/golden are the reference's own test images (test-data/1.jpg, 2.jpg, used by
akaze/tests/integration-test.rs)."""
import json
import os

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def photo():
    return np.asarray(PIL.open(os.path.join(GOLDEN, "1.jpg")).convert("RGB"))[200:533, 300:817].copy()


JPEG_VARIANTS = {
    "444": dict(quality=90, subsampling=0), "422": dict(quality=90, subsampling=1),
    "420": dict(quality=90, subsampling=2), "q50": dict(quality=50, subsampling=2),
    "progressive_420": dict(quality=85, subsampling=2, progressive=True),
    "progressive_444": dict(quality=85, subsampling=0, progressive=True),
    "optimised_tables": dict(quality=85, subsampling=2, optimize=True),
    "restart_blocks": dict(quality=85, subsampling=2, restart_marker_blocks=7),
    "restart_rows": dict(quality=85, subsampling=2, restart_marker_rows=1),
}


@pytest.mark.parametrize("name", sorted(JPEG_VARIANTS))
def test_jpeg_decode_close_to_libjpeg(amd, photo, tmp_path, name):
    p = str(tmp_path / (name + ".jpg"))
    PIL.fromarray(photo).save(p, **JPEG_VARIANTS[name])
    got = amd.load_image(p)
    exp = np.asarray(PIL.open(p).convert("RGB"))
    assert got.shape == exp.shape == photo.shape
    d = np.abs(got.astype(int) - exp.astype(int))
    assert d.max() <= 4 and d.mean() < 0.2, (d.max(), d.mean())


def test_jpeg_gray_odd_and_tiny_sizes(amd, photo, tmp_path):
    for name, arr, kw in (("gray", photo[..., 1], {}), ("gray_prog", photo[..., 1], dict(progressive=True)),
                          ("odd", photo[:101, :77], dict(subsampling=2)), ("tiny", photo[:5, :3], dict(subsampling=2)),
                          ("one", photo[:1, :1], dict(subsampling=2))):
        p = str(tmp_path / (name + ".jpg"))
        PIL.fromarray(arr).save(p, quality=88, **kw)
        got = amd.load_image(p)
        exp = np.asarray(PIL.open(p).convert("L" if arr.ndim == 2 else "RGB"))
        assert got.shape == exp.shape
        assert np.abs(got.astype(int) - exp.astype(int)).max() <= 4


def test_reference_test_images_decode_and_luma(amd):
    for n in ("1.jpg", "2.jpg"):
        p = os.path.join(GOLDEN, n)
        rgb, luma = amd.load_image_rgb(p), amd.load_image_luma(p)
        assert rgb.shape == (1512, 2016, 3) and luma.shape == (1512, 2016)
        exp = np.asarray(PIL.open(p).convert("RGB"))
        assert np.abs(rgb.astype(int) - exp.astype(int)).max() <= 4
        # DynamicImage::to_luma (believed): f32 weights, truncating cast
        w = (np.float32(0.2126) * rgb[..., 0].astype(np.float32) + np.float32(0.7152) * rgb[..., 1].astype(np.float32)
             + np.float32(0.0722) * rgb[..., 2].astype(np.float32))
        assert np.abs(luma.astype(int) - w.astype(np.uint8).astype(int)).max() <= 1  # summation order of the f32 products


def test_lossless_formats_decode_exactly(amd, photo, tmp_path):
    im = PIL.fromarray(photo)
    cases = [("rgb.png", im, "RGB"), ("gray.png", im.convert("L"), "L"), ("rgba.png", im.convert("RGBA"), "RGB"),
             ("palette.png", im.convert("P"), "RGB"), ("gray_alpha.png", im.convert("LA"), "L"),
             ("bilevel.png", im.convert("1"), "L"), ("gray.pgm", im.convert("L"), "L"), ("rgb.ppm", im, "RGB")]
    for name, img, mode in cases:
        p = str(tmp_path / name)
        img.save(p)
        got = amd.load_image(p)
        exp = np.asarray(PIL.open(p).convert(mode))
        assert got.shape == exp.shape and np.array_equal(got, exp), name
    # a luma file passes through to_luma unchanged; to_rgb replicates it
    p = str(tmp_path / "gray.png")
    assert np.array_equal(amd.load_image_luma(p), np.asarray(im.convert("L")))
    assert np.array_equal(amd.load_image_rgb(p)[..., 2], np.asarray(im.convert("L")))


def test_png_writer_round_trips(amd, photo, tmp_path):
    for arr in (photo, photo[..., 0].copy()):
        p = str(tmp_path / "out.png")
        amd.save_png(p, arr)
        assert np.array_equal(np.asarray(PIL.open(p)), arr)
        assert np.array_equal(amd.load_image(p), arr)


def test_image_errors(amd, tmp_path):
    with pytest.raises(amd.AkazeError) as e:
        amd.load_image(str(tmp_path / "missing.jpg"))
    assert e.value.status == -8
    bad = tmp_path / "garbage.bin"
    bad.write_bytes(b"not an image at all")
    with pytest.raises(amd.AkazeError) as e:
        amd.load_image(str(bad))
    assert e.value.status == -6
    trunc = tmp_path / "truncated.png"
    good = tmp_path / "good.png"
    PIL.fromarray(np.zeros((8, 8), np.uint8)).save(str(good))
    trunc.write_bytes(good.read_bytes()[:40])
    with pytest.raises(amd.AkazeError):
        amd.load_image(str(trunc))


def test_c1_reference_jpeg_through_the_cpu_path(amd, ref):
    """BASELINE.json configs[0] / akaze/tests/integration-test.rs:41-70: test-data/1.jpg with Config::default()
    through the CPU path end to end (here: this repo's decoder + the oracle) — plumbing, no GPU."""
    luma = amd.load_image_luma(os.path.join(GOLDEN, "1.jpg"))
    r = ref.extract(luma, threads=8)
    assert r.num_levels == 16 and r.num_keypoints > 100 and r.desc_bytes == 61
    assert r.descriptors().shape == (r.num_keypoints, 61)
    kp = r.keypoints()
    assert kp["x"].min() >= 0 and kp["x"].max() < 2016 and kp["y"].max() < 1512


def test_config_json_matches_serde_layout(amd):
    text = amd.config_to_json()
    assert text == ('{"num_sublevels":4,"max_octave_evolution":4,"base_scale_offset":1.6,"initial_contrast":0.001,'
                    '"contrast_percentile":0.7,"contrast_factor_num_bins":300,"derivative_factor":1.5,'
                    '"detector_threshold":0.001,"descriptor_channels":3,"descriptor_pattern_size":10}')
    assert json.loads(text)["contrast_factor_num_bins"] == 300
    cfg = amd.config_from_json('{ "num_sublevels": 5, "detector_threshold": 0.0005,\n "descriptor_channels": 1, "unknown": [1, {"a": 2}] }')
    assert (cfg.num_sublevels, cfg.detector_threshold, cfg.descriptor_channels, cfg.max_octave_evolution) == (5, 0.0005, 1, 4)
    assert json.loads(amd.config_to_json(cfg))["num_sublevels"] == 5
    with pytest.raises(amd.AkazeError):
        amd.config_from_json('{"num_sublevels": ')


def test_save_plane_normalises_like_the_reference(amd, tmp_path):
    """types::image::save = normalize (min/max) -> (v * 255) as u8 (image.rs:150-197)."""
    rng = np.random.default_rng(5)
    plane = (rng.standard_normal((37, 53)) * 3).astype(np.float32)
    p = str(tmp_path / "plane.png")
    amd.save_plane_png(p, plane)
    mn, mx = plane.min(), plane.max()
    exp = (((plane - mn) / np.float32(mx - mn)) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(np.asarray(PIL.open(p)), exp)


def _blend(img, x, y, col):
    img[y, x] = ((col.astype(np.float32) + img[y, x].astype(np.float32)) / np.float32(2)).astype(np.uint8)


def _circle(img, px, py, col, radius):  # types/image.rs:417-444, out-of-image pixels skipped
    h, w = img.shape[:2]
    cx, cy, r = int(px), int(py), int(radius)
    for x in range(max(cx - r, 0), cx + r):
        for y in range(max(cy - r, 0), cy + r):
            if x < w and y < h and np.sqrt(np.float32(x - px) ** 2 + np.float32(y - py) ** 2) <= radius:
                _blend(img, x, y, col)


def test_draw_keypoints_and_matches(amd):
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (60, 80, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (50, 70, 3), dtype=np.uint8)
    kp0 = np.zeros(3, amd.KEYPOINT_DTYPE)
    kp0["x"], kp0["y"], kp0["size"] = [20.5, 40.25, 78.0], [30.0, 10.75, 58.0], [4.8, 7.2, 3.0]
    kp1 = np.zeros(2, amd.KEYPOINT_DTYPE)
    kp1["x"], kp1["y"] = [10.0, 60.0], [20.0, 40.0]
    # colours: three u8 reads per keypoint from the thread's persistent Xorshift128+ source (random_color, image.rs:385-392)
    from test_reference_outputs import xorshift128plus
    amd.random_seed(42, 69)
    drawn = amd.draw_keypoints(a, kp0)
    stream = xorshift128plus()
    exp = a.copy()
    for k in kp0:
        col = np.array([next(stream) & 0xFF for _ in range(3)], np.uint8)
        _circle(exp, np.float32(k["x"]), np.float32(k["y"]), col, np.float32(k["size"]))
    assert np.array_equal(drawn, exp) and not np.array_equal(drawn, a)
    again = amd.draw_keypoints(a, kp0)            # the stream continues: other colours
    assert not np.array_equal(again, drawn)
    amd.random_seed(42, 69)
    assert np.array_equal(amd.draw_keypoints(a, kp0), drawn)
    m = np.zeros(2, amd.MATCH_DTYPE)
    m["index_0"], m["index_1"] = [0, 1], [1, 0]
    out = amd.draw_matches(a, b, kp0, kp1, m)
    assert out.shape == (60, 160, 3)                      # two halves, each as wide as the wider image
    assert np.array_equal(out[:, :80], a) and np.array_equal(out[:50, 80:150], b)
    assert not out[50:, 80:].any() and not out[:, 150:].any()   # outside the second image: black
    # line radius = combined_height / 500 (feature_match.rs:78) truncates to 0 pixels below 500 rows: nothing is drawn
    # on small images, as in the reference; on a tall pair the lines appear
    ta, tb = np.repeat(a, 10, axis=0), np.repeat(b, 10, axis=0)
    tall = amd.draw_matches(ta, tb, kp0, kp1, m)
    assert tall.shape == (600, 160, 3)
    changed = np.any(tall[:, :80] != ta, axis=2)
    assert 10 < changed.sum() < 2000                       # thin lines were blended into the first half
    with pytest.raises(amd.AkazeError):
        m["index_1"] = [5, 0]
        amd.draw_matches(a, b, kp0, kp1, m)
