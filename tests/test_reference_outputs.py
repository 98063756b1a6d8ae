"""Pinning against the reference's OWN published outputs: test-data/keypoints-1.jpg and keypoints-2.jpg are what the
reference crate drew for test-data/1.jpg and 2.jpg (`extract_features -d`: draw_keypoints_to_image on the input image,
akaze-util/src/bin/extract_features.rs:94-103); all four files are committed under tests/golden.

The drawing blends, for keypoint i of the returned list, a disc of radius keypoint.size at keypoint.point in the colour
random_color() returns at that moment (types/image.rs:385-444) — three u8 reads from the `random` crate's thread-local
default source.  So the picture carries, per keypoint, its POSITION, its SIZE and — through the colour — its INDEX in the
list.  Checked here, for the CPU oracle run on this repo's decode of the same JPEG:

  * the union of the oracle's keypoint discs covers the changed pixels of the published picture (IoU > 0.95);
  * for every keypoint whose disc does not overlap another one, the disc colour recovered from the picture equals
    values 3i, 3i+1, 3i+2 of Xorshift128+ seeded [42, 69] at the keypoint's index i in the ORACLE's list.

The second property only holds if the oracle returns the same number of keypoints in the same order, at the same places
and sizes, as the reference did (one missing or extra keypoint would shift the colour of every later one).  It also pins
the generator, its seed and the fact that `random::default()` is one persistent stream per thread — which is what the
RANSAC sampling consumes (estimate_fundamental_matrix.rs:118-121).  No GPU needed; the GPU path is then tied to the
oracle bit for bit by tests/test_gpu_*.py."""
import os

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
M64 = (1 << 64) - 1


def luma_pil(path):
    """image::open(path).to_luma() with an INDEPENDENT decoder (PIL / libjpeg-turbo instead of this repository's
    akz_image.cpp): `image` 0.21 weights in f32, truncating cast."""
    rgb = np.asarray(PIL.open(path).convert("RGB")).astype(np.float32)
    l = (np.float32(0.2126) * rgb[..., 0] + np.float32(0.7152) * rgb[..., 1]) + np.float32(0.0722) * rgb[..., 2]
    return l.astype(np.uint8)


_EXTRACTED = {}


def extracted(ref, name, decoder, amd=None):
    """the oracle's result for tests/golden/<name>.jpg decoded by `decoder` ("product" or "pil"), computed once"""
    key = (name, decoder)
    if key not in _EXTRACTED:
        src = os.path.join(GOLDEN, f"{name}.jpg")
        luma = amd.load_image_luma(src) if decoder == "product" else luma_pil(src)
        _EXTRACTED[key] = ref.extract(luma, threads=8)
    return _EXTRACTED[key]


def xorshift128plus(s0=42, s1=69):
    while True:
        x, y = s0, s1
        s0 = y
        x ^= (x << 23) & M64
        x ^= x >> 17
        x ^= y ^ (y >> 26)
        s1 = x
        yield (x + y) & M64


def disc_cover(kp, shape):
    """coverage count and last-writer index per pixel, with draw_circle's integer window (types/image.rs:417-444)"""
    h, w = shape
    cnt = np.zeros((h, w), np.int32)
    owner = np.full((h, w), -1, np.int32)
    yy, xx = np.mgrid[0:h, 0:w]
    for i, k in enumerate(kp):
        cx, cy, rad = np.float32(k["x"]), np.float32(k["y"]), np.float32(k["size"])
        x0, x1 = max(int(cx) - int(rad), 0), min(int(cx) + int(rad), w)
        y0, y1 = max(int(cy) - int(rad), 0), min(int(cy) + int(rad), h)
        inside = (xx[y0:y1, x0:x1] - cx) ** 2 + (yy[y0:y1, x0:x1] - cy) ** 2 <= rad * rad
        cnt[y0:y1, x0:x1] += inside
        owner[y0:y1, x0:x1][inside] = i
    return cnt, owner


def own_pixels(kp, i, single, owner, published, base):
    """pixels of keypoint i's disc that no other disc touches: (published, base) values, from the disc's window only"""
    h, w = single.shape
    cx, cy, rad = int(kp["x"][i]), int(kp["y"][i]), int(kp["size"][i]) + 1
    x0, x1, y0, y1 = max(cx - rad, 0), min(cx + rad + 1, w), max(cy - rad, 0), min(cy + rad + 1, h)
    sel = single[y0:y1, x0:x1] & (owner[y0:y1, x0:x1] == i)
    return published[y0:y1, x0:x1][sel], base[y0:y1, x0:x1][sel]


@pytest.mark.parametrize("name", ["1", "2"])
def test_oracle_keypoints_match_the_reference_picture(amd, ref, name):
    src = os.path.join(GOLDEN, f"{name}.jpg")
    base = np.asarray(PIL.open(src).convert("RGB")).astype(np.int32)
    published = np.asarray(PIL.open(os.path.join(GOLDEN, f"keypoints-{name}.jpg")).convert("RGB")).astype(np.int32)
    assert base.shape == published.shape == (1512, 2016, 3)
    kp = extracted(ref, name, "product", amd).keypoints()
    assert len(kp) > 3000
    cnt, owner = disc_cover(kp, base.shape[:2])

    changed = np.abs(published - base).sum(axis=2) > 40       # JPEG noise of the re-encoded picture stays below that
    mine = cnt > 0
    iou = (mine & changed).sum() / (mine | changed).sum()
    recall = (mine & changed).sum() / changed.sum()
    assert iou > 0.95 and recall > 0.99, (iou, recall)

    stream = xorshift128plus()
    vals = np.array([next(stream) & 0xFF for _ in range(3 * len(kp))], np.float64).reshape(-1, 3)
    single = cnt == 1
    checked = good = 0
    last_checked = 0
    for i in range(len(kp)):
        pub_px, base_px = own_pixels(kp, i, single, owner, published, base)
        if len(pub_px) < 30:
            continue
        colour = np.median(2 * pub_px - base_px, axis=0)   # blend: out = (colour + pixel) / 2
        checked += 1
        last_checked = i
        good += np.abs(np.clip(colour, 0, 255) - vals[i]).max() < 14
    assert checked > 800 and last_checked > 0.98 * len(kp)
    assert good / checked > 0.985, (good, checked)


def test_product_drawing_reproduces_the_reference_picture(amd, ref):
    """akz_draw_keypoints on the input image with the oracle's keypoints, from a freshly seeded source, is the published
    picture up to its JPEG re-encoding."""
    src = os.path.join(GOLDEN, "1.jpg")
    kp = extracted(ref, "1", "product", amd).keypoints()
    amd.random_seed(42, 69)
    drawn = amd.draw_keypoints(amd.load_image_rgb(src), kp).astype(np.int32)
    published = np.asarray(PIL.open(os.path.join(GOLDEN, "keypoints-1.jpg")).convert("RGB")).astype(np.int32)
    d = np.abs(drawn - published)
    assert d.mean() < 3.0 and (d.max(axis=2) > 48).mean() < 0.01, (d.mean(), (d.max(axis=2) > 48).mean())
    # the same call without reseeding continues the stream: other colours
    again = amd.draw_keypoints(amd.load_image_rgb(src), kp).astype(np.int32)
    assert np.abs(again - published).mean() > 2 * d.mean()


def test_oracle_matches_are_the_lines_of_the_reference_match_picture(amd, ref):
    """test-data/match_image.jpg is the reference's draw_matches output for 1.jpg / 2.jpg (feature_match.rs:32-82, written
    by extract_and_match -m or the integration test): the two images side by side and one line per match that survived
    RANSAC, from keypoint_0.point to keypoint_1.point shifted by the first image's width.  The oracle's descriptor_match
    + remove_outliers on the same images must select (nearly) those pairs: a line is drawn along >= 97 % of the oracle's
    inlier matches, while random keypoint pairs rarely lie on one (< 15 %; measured 7.5 %).  (RANSAC is not bit-reproducible even
    reference-to-reference — HashSet order, SVD rounding — so the inlier sets may differ by a few matches.)  This ties
    the descriptors and the Hamming matcher of the oracle to the reference's real output: wrong descriptor bits would
    pair other keypoints."""
    paths = [os.path.join(GOLDEN, n) for n in ("1.jpg", "2.jpg")]
    base = np.concatenate([np.asarray(PIL.open(p).convert("RGB")).astype(np.int32) for p in paths], axis=1)
    published = np.asarray(PIL.open(os.path.join(GOLDEN, "match_image.jpg")).convert("RGB")).astype(np.int32)
    assert published.shape == base.shape == (1512, 4032, 3)
    changed = np.abs(published - base).sum(axis=2) > 40
    r0, r1 = (extracted(ref, n, "product", amd) for n in ("1", "2"))
    k0, k1 = r0.keypoints(), r1.keypoints()
    matches = ref.descriptor_match(r0.descriptors(), r1.descriptors(), 10000, 0.86)
    ref.random_seed(42, 69)
    inliers = ref.remove_outliers(k0, k1, matches, 1000, 0.05, 3.0)
    assert len(matches) > 300 and 0.4 * len(matches) < len(inliers) < len(matches)
    h, w = changed.shape

    def drawn(i0, i1):
        x0, y0, x1, y1 = float(k0["x"][i0]), float(k0["y"][i0]), float(k1["x"][i1]) + w / 2, float(k1["y"][i1])
        n = int(max(abs(x1 - x0), abs(y1 - y0), 2))
        xs = np.linspace(x0, x1, n).astype(int).clip(0, w - 1)
        ys = np.linspace(y0, y1, n).astype(int).clip(0, h - 1)
        return changed[ys, xs].mean() > 0.9

    on_line = np.array([drawn(int(m["index_0"]), int(m["index_1"])) for m in inliers])
    assert on_line.mean() > 0.97, on_line.mean()
    all_on_line = np.array([drawn(int(m["index_0"]), int(m["index_1"])) for m in matches])
    assert 0.9 * len(inliers) < all_on_line.sum() < 1.2 * len(inliers)       # the reference kept about as many
    rng = np.random.default_rng(3)
    control = np.array([drawn(int(a), int(b)) for a, b in zip(rng.integers(0, len(k0), 400), rng.integers(0, len(k1), 400))])
    assert control.mean() < 0.15, control.mean()   # the lines are nearly parallel, so a few random pairs fall onto one


@pytest.mark.parametrize("name", ["1", "2"])
def test_oracle_pin_holds_with_an_independent_jpeg_decoder(ref, name):
    """The same colour-index pin with NO product code on the path: the JPEG is decoded by PIL.  libjpeg-turbo and the
    reference's jpeg-decoder differ by up to 2 levels on 0.15 % of the pixels, which adds or removes a handful of
    keypoints near the detector threshold (7395 instead of 7393 on 1.jpg), so the oracle's index may run ahead of or
    behind the reference's by a few places — but it must do so as a slowly changing offset: > 98.5 % of the isolated
    discs carry the stream colour of index i + d with |d| <= 6, and d changes only a few dozen times over the ~1 500
    checked discs (measured: 19 changes on 1.jpg, final offset -2)."""
    src = os.path.join(GOLDEN, f"{name}.jpg")
    base = np.asarray(PIL.open(src).convert("RGB")).astype(np.int32)
    published = np.asarray(PIL.open(os.path.join(GOLDEN, f"keypoints-{name}.jpg")).convert("RGB")).astype(np.int32)
    kp = extracted(ref, name, "pil").keypoints()
    assert len(kp) > 3000
    cnt, owner = disc_cover(kp, base.shape[:2])
    changed = np.abs(published - base).sum(axis=2) > 40
    mine = cnt > 0
    assert (mine & changed).sum() / (mine | changed).sum() > 0.94
    stream = xorshift128plus()
    vals = np.array([next(stream) & 0xFF for _ in range(3 * (len(kp) + 16))], np.float64).reshape(-1, 3)
    single = cnt == 1
    checked = good = 0
    drifts = []
    for i in range(len(kp)):
        pub_px, base_px = own_pixels(kp, i, single, owner, published, base)
        if len(pub_px) < 30:
            continue
        colour = np.clip(np.median(2 * pub_px - base_px, axis=0), 0, 255)
        checked += 1
        cands = [d for d in range(-6, 7) if 0 <= i + d < len(vals) and np.abs(colour - vals[i + d]).max() < 14]
        if cands:
            good += 1
            prev = drifts[-1] if drifts else 0
            drifts.append(min(cands, key=lambda d: abs(d - prev)))
    assert checked > 800 and good / checked > 0.985, (good, checked)
    changes = int(np.count_nonzero(np.diff(drifts)))
    assert changes <= 40 and abs(drifts[-1]) <= 6, (changes, drifts[-1])


def test_every_line_of_the_reference_match_picture_is_an_oracle_nearest_neighbour_pair(ref):
    """The converse of the test above, per line and with no product code on the path (PIL decode): rasterise the lines
    (i, nn(i)) of the oracle's ratio-test matches that the picture shows, and require that they EXPLAIN the picture —
    more than 99 % of all changed pixels of match_image.jpg lie on such a line.  A drawn line whose endpoints the
    oracle's Hamming search does not pair (wrong descriptor bits on either side) would leave its pixels unexplained;
    with half of the explained lines removed the coverage drops well below the bar (control)."""
    paths = [os.path.join(GOLDEN, n) for n in ("1.jpg", "2.jpg")]
    base = np.concatenate([np.asarray(PIL.open(p).convert("RGB")).astype(np.int32) for p in paths], axis=1)
    published = np.asarray(PIL.open(os.path.join(GOLDEN, "match_image.jpg")).convert("RGB")).astype(np.int32)
    changed = np.abs(published - base).sum(axis=2) > 40
    h, w = changed.shape
    r0, r1 = (extracted(ref, n, "pil") for n in ("1", "2"))
    k0, k1 = r0.keypoints(), r1.keypoints()
    matches = ref.descriptor_match(r0.descriptors(), r1.descriptors(), 10000, 0.86)

    def samples(i0, i1):
        x0, y0, x1, y1 = float(k0["x"][i0]), float(k0["y"][i0]), float(k1["x"][i1]) + w / 2, float(k1["y"][i1])
        n = int(max(abs(x1 - x0), abs(y1 - y0), 2))
        return np.linspace(x0, x1, n).astype(int).clip(0, w - 1), np.linspace(y0, y1, n).astype(int).clip(0, h - 1)

    shown = []
    for m in matches:
        xs, ys = samples(int(m["index_0"]), int(m["index_1"]))
        if changed[ys, xs].mean() > 0.9:
            shown.append((xs, ys))
    assert len(shown) > 250
    rad = h // 500 + 3  # draw_line's radius (feature_match.rs: height / 500) plus JPEG blur
    disc = [(dx, dy) for dx in range(-rad, rad + 1) for dy in range(-rad, rad + 1) if dx * dx + dy * dy <= rad * rad]

    def coverage(lines):
        mask = np.zeros_like(changed)
        for xs, ys in lines:
            for dx, dy in disc:
                mask[(ys + dy).clip(0, h - 1), (xs + dx).clip(0, w - 1)] = True
        return (changed & mask).sum() / changed.sum()

    full = coverage(shown)
    assert full > 0.99, full
    assert coverage(shown[::2]) < 0.93   # control: the check notices missing lines
