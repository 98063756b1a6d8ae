"""GPU parity tests, one per op of the C ABI, against the CPU oracle on the same seeded inputs.
Bar: bit-exact (f32 results compared by value so that -0.0 == +0.0, NaNs never expected)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    import torch
    torch.cuda.synchronize()
    return t.cpu().numpy()


def same(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert not np.isnan(a).any() and not np.isnan(b).any()
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError(f"{len(bad)} of {a.size} values differ, first at {bad[0]}: {a[tuple(bad[0])]!r} vs "
                             f"{b[tuple(bad[0])]!r}; max abs diff {np.abs(a - b).max()}")


def rand_img(h, w, seed, lo=0.0, hi=1.0):
    return np.random.default_rng(seed).uniform(lo, hi, (h, w)).astype(np.float32)


SHAPES = [(37, 53), (64, 64), (135, 240), (101, 259)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("klen", [3, 5, 9, 13])
def test_horizontal_vertical_filter(ctx, ref, shape, klen):
    img = rand_img(*shape, seed=klen)
    kern = np.random.default_rng(klen + 1).standard_normal(klen).astype(np.float32)
    same(host(ctx.horizontal_filter(dev(img), kern)), ref.horizontal_filter(img, kern))
    same(host(ctx.vertical_filter(dev(img), kern)), ref.vertical_filter(img, kern))


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("sigma", [1.0, 1.6, 2.5])
def test_gaussian_blur(ctx, ref, shape, sigma):
    img = rand_img(*shape, seed=2)
    same(host(ctx.gaussian_blur(dev(img), sigma)), ref.gaussian_blur(img, sigma))


def test_gaussian_blur_u8_unit_float(ctx, ref):
    u8 = np.random.default_rng(0).integers(0, 256, (77, 131), dtype=np.uint8)
    unit = (u8.astype(np.float32) * np.float32(1.0)) / np.float32(255.0)  # image.rs:136
    same(host(ctx.gaussian_blur(dev(u8), 1.6)), ref.gaussian_blur(unit, 1.6))


def test_batch_of_planes(ctx, ref):
    imgs = np.stack([rand_img(45, 70, s) for s in range(3)])
    out = host(ctx.gaussian_blur(dev(imgs), 1.0))
    for i in range(3):
        same(out[i], ref.gaussian_blur(imgs[i], 1.0))


@pytest.mark.parametrize("shape", [(37, 53), (64, 64), (135, 241), (2, 2)])
def test_half_size(ctx, ref, shape):
    img = rand_img(*shape, seed=3)
    same(host(ctx.half_size(dev(img))), ref.half_size(img))


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("sigma", [1, 2, 3, 4])
def test_scharr(ctx, ref, shape, sigma):
    img = rand_img(*shape, seed=4, lo=-1.0)
    same(host(ctx.scharr(dev(img), True, False, sigma)), ref.scharr(img, True, False, sigma))
    same(host(ctx.scharr(dev(img), False, True, sigma)), ref.scharr(img, False, True, sigma))


@pytest.mark.parametrize("shape", [(20, 20), (63, 130), (257, 33)])
@pytest.mark.parametrize("sigma", [1, 2, 4])
def test_scharr_both_and_neither_order(ctx, ref, shape, sigma):
    """derivatives.rs:118-122 (both: horizontal + horizontal) and :127-128 (neither: zeros), also on a batch"""
    img = rand_img(*shape, seed=14, lo=-1.0)
    both = ref.scharr(img, True, True, sigma)
    assert np.array_equal(both, ref.scharr(img, True, False, sigma) * np.float32(2))  # what the reference computes
    same(host(ctx.scharr(dev(img), True, True, sigma)), both)
    none = host(ctx.scharr(dev(img), False, False, sigma))
    same(none, ref.scharr(img, False, False, sigma))
    assert none.shape == img.shape and not none.any()
    same(host(ctx.scharr(dev(img), False, False, 0)), none)  # sigma_size is not looked at without an order
    batch = np.stack([img, rand_img(*shape, seed=15, lo=-1.0)])
    got = host(ctx.scharr(dev(batch), True, True, sigma))
    for i in range(2):
        same(got[i], ref.scharr(batch[i], True, True, sigma))


@pytest.mark.parametrize("k", [0.0043, 0.03, 0.5])
def test_pm_g2(ctx, ref, k):
    lx, ly = rand_img(60, 90, 5, -0.2, 0.2), rand_img(60, 90, 6, -0.2, 0.2)
    same(host(ctx.pm_g2(dev(lx), dev(ly), k)), ref.pm_g2(lx, ly, k))


def test_pm_g2_reciprocal_boundaries(ctx):
    """pm_g2's `(1.0 / x) as f32` (lib.rs:36) takes the refined hardware reciprocal except where the f32 rounding could
    depend on the last bits of the f64 quotient (akz_pm_g2.hpp).  Doubles whose quotient lies ON an f32 rounding boundary,
    a few ulp either side of one, and far from any, against numpy's correctly rounded division."""
    import torch
    rng = np.random.default_rng(21)
    f = rng.uniform(2.0 ** -20, 1.0, 200_000).astype(np.float32)
    mid = (f.astype(np.float64) + np.nextafter(f, np.float32(2)).astype(np.float64)) * 0.5  # boundaries in (0, 1]
    x0 = 1.0 / mid
    xs = [x0]
    for d in (1, 2, 3, 5, 9, 17, 40):  # quotients within a few ulp(f64) of the boundary, both sides
        for sgn in (1, -1):
            x = x0.copy()
            for _ in range(d):
                x = np.nextafter(x, np.inf if sgn > 0 else -np.inf)
            xs.append(x)
    xs.append(1.0 + rng.uniform(0, 1, 2_000_000) * 10.0 ** rng.uniform(-6, 6, 2_000_000))  # what pm_g2 sees
    xs.append(np.array([1.0, 2.0, 4.0, 3.0, 1e19, 2.0 ** 64, 2.0 ** 100, 1e300, 2.0 ** 127, 2.0 ** 140, 2.0 ** 149,
                        2.0 ** 150, np.inf, 0.5, 1e-30]))
    x = np.maximum(np.concatenate(xs), 1e-30)
    got = host(ctx.debug_rcp_f64_to_f32(torch.from_numpy(x).cuda()))
    with np.errstate(over="ignore", under="ignore"):
        want = (1.0 / x).astype(np.float32)
    assert np.array_equal(got, want), (x[got != want][:5], got[got != want][:5], want[got != want][:5])
    # the boundary cases must really be there: without the guard a plain (float)r would differ somewhere
    assert (np.abs((1.0 / x0) - mid) <= np.spacing(mid) * 2).mean() > 0.9


@pytest.mark.parametrize("shape", SHAPES)
def test_flow_is_scharr1_plus_pm_g2(ctx, ref, shape):
    ls = rand_img(*shape, seed=7)
    k = 0.0123
    exp = ref.pm_g2(ref.scharr(ls, True, False, 1), ref.scharr(ls, False, True, 1), k)
    same(host(ctx.flow(dev(ls), k)), exp)
    # octave scaling: k * 0.75 * 0.75 multiplied step by step in f64 (lib.rs:84)
    exp2 = ref.pm_g2(ref.scharr(ls, True, False, 1), ref.scharr(ls, False, True, 1), (k * 0.75) * 0.75)
    same(host(ctx.flow(dev(ls), k, 2)), exp2)


@pytest.mark.parametrize("shape", SHAPES + [(270, 480)])
def test_contrast_factor(ctx, ref, shape):
    img = ref.gaussian_blur(rand_img(*shape, seed=8), 1.6)
    got = float(host(ctx.contrast_factor(dev(img), 0.7, 1.0, 300))[0])
    assert got == ref.contrast_factor(img, 0.7, 1.0, 300)


def test_contrast_factor_flat_image(ctx, ref):
    img = np.full((40, 50), 0.25, np.float32)
    assert float(host(ctx.contrast_factor(dev(img)))[0]) == ref.contrast_factor(img)


FED_TAUS = np.array([0.19623365730888334, 3.7012260958150769, 0.12604081556156974, 0.25, 1.5, 0.07, 2.2, 0.4,
                     0.9, 0.11, 3.1, 0.6, 0.33])


@pytest.mark.parametrize("mode", [2, 0])  # 2: k_fed_own (<= 8 steps per launch, <= 16 for small launches), 0: k_fed_step
@pytest.mark.parametrize("shape", SHAPES + [(3, 3), (5, 64), (64, 5), (33, 130), (270, 480)])
@pytest.mark.parametrize("ntau", [1, 2, 4, 5, 8, 13])
def test_fed_steps_all_border_cases(ctx, ref, shape, ntau, mode):
    lt = rand_img(*shape, seed=9)
    c = rand_img(*shape, seed=10)
    taus = FED_TAUS[:ntau]
    exp, step = lt.copy(), None
    for t in taus:
        exp, step = ref.fed_step(exp, c, t)
    ctx.set_fed_mode(mode)
    try:
        d_lt = dev(lt)
        d_step = ctx.fed_steps(d_lt, dev(c), taus, want_lstep=True)
        same(host(d_lt), exp)
        same(host(d_step), step)
    finally:
        ctx.set_fed_mode(2)


def test_fed_fused_batch_and_unaligned_width(ctx, ref):
    """Batch of planes whose width is not a multiple of 4 (scalar load/store path of k_fed_own)."""
    lt = np.stack([rand_img(70, 131, s) for s in range(3)])
    c = np.stack([rand_img(70, 131, 10 + s) for s in range(3)])
    taus = FED_TAUS[:7]
    d_lt = dev(lt)
    ctx.fed_steps(d_lt, dev(c), taus)
    got = host(d_lt)
    for i in range(3):
        exp = lt[i].copy()
        for t in taus:
            exp, _ = ref.fed_step(exp, c[i], t)
        same(got[i], exp)


@pytest.mark.parametrize("sigma", [2, 3, 4])
def test_detector_response(ctx, ref, sigma):
    ls = rand_img(96, 130, seed=11)
    got = ctx.detector_response(dev(ls), sigma)
    lx = ref.scharr(ls, True, False, sigma)
    ly = ref.scharr(ls, False, True, sigma)
    lxx = ref.scharr(lx, True, False, sigma)
    lyy = ref.scharr(ly, False, True, sigma)
    lxy = ref.scharr(lx, False, True, sigma)
    q = np.float32(sigma ** 4)
    ldet = ((lxx * lyy) - (lxy * lxy)) * q
    for name, exp in (("Lx", lx), ("Ly", ly), ("Lxx", lxx), ("Lyy", lyy), ("Lxy", lxy), ("Ldet", ldet)):
        same(host(got[name]), exp)
    lean = ctx.detector_response(dev(ls), sigma, keep_second=False)
    same(host(lean["Ldet"]), ldet)


@pytest.fixture(params=[0, 1, 3], ids=["popcount", "mfma_i8", "mfma_fp4"])
def mctx(request, ctx):
    """The context with the matcher kernel forced: 0 = popcount scan, 1 = matrix-core scan on int8 operands, 3 = on FP4 operands."""
    ctx.set_match_mode(request.param)
    yield ctx
    ctx.set_match_mode(2)


def test_descriptor_match(mctx, ref):
    ctx = mctx
    rng = np.random.default_rng(12)
    d0 = rng.integers(0, 256, (300, 61), dtype=np.uint8)
    d1 = rng.integers(0, 256, (517, 61), dtype=np.uint8)
    for i in range(0, 300, 3):  # plant near-duplicates so that the ratio test passes
        d1[(i * 7) % 517] = d0[i]
        d1[(i * 7) % 517, i % 61] ^= 0x11
    d1[100] = d0[4]; d1[400] = d0[4]  # exact tie
    for ratio, thr in ((0.86, 10000), (0.5, 10000), (0.99, 40)):
        got = ctx.descriptor_match(d0, d1, thr, ratio)
        exp = ref.descriptor_match(d0, d1, thr, ratio)
        assert len(got) == len(exp) and len(got) > 0
        assert np.array_equal(got, exp)
    assert len(ctx.descriptor_match(d0, np.zeros((0, 61), np.uint8))) == 0
    assert np.array_equal(ctx.descriptor_match(d0[:5], d1[:1]), ref.descriptor_match(d0[:5], d1[:1]))
    assert len(ctx.descriptor_match(np.zeros((0, 61), np.uint8), d1)) == 0


@pytest.mark.parametrize("n0,n1", [(1, 1), (2, 33), (33, 31), (64, 64), (513, 1000), (700, 63), (1025, 4097)])
def test_descriptor_match_ragged_sizes(ctx, ref, n0, n1):
    """Set sizes around the tile sizes of the matrix-core matcher (32 train rows, 512 queries per workgroup), low-entropy
    descriptors with many exact ties and duplicates: both matcher kernels against the oracle's sequential scan."""
    rng = np.random.default_rng(n0 * 7919 + n1)
    base = rng.integers(0, 256, (8, 61), dtype=np.uint8)
    d0 = base[rng.integers(0, 8, n0)].copy()
    d1 = base[rng.integers(0, 8, n1)].copy()
    d0[rng.random(d0.shape) < 0.02] ^= 0x10  # a few differing bits: many equal distances, some exact duplicates
    d1[rng.random(d1.shape) < 0.02] ^= 0x01
    for ratio, thr in ((0.86, 10000), (1.5, 10000), (0.99, 12)):
        exp = ref.descriptor_match(d0, d1, thr, ratio)
        for mode in (1, 0, 3):
            ctx.set_match_mode(mode)
            try:
                got = ctx.descriptor_match(d0, d1, thr, ratio)
            finally:
                ctx.set_match_mode(2)
            assert np.array_equal(got, exp), (mode, ratio, thr, len(got), len(exp))


@pytest.mark.parametrize("chunks_per_set", [None, 2, 3, 16])
def test_descriptor_match_sets_chunked(ctx, ref, chunks_per_set):
    """A multi-set launch with every set cut into several chunks (the form large sets take so that the workgroups fill
    whole rounds of the chip; forced here through akz_debug_set_match_chunks): sets shorter than the chunk count leave chunks
    empty, equal minima in different chunks must resolve to the lowest row, and the per-set pruning bound is shared by
    the set's chunks."""
    import torch
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, (24, 61), dtype=np.uint8)

    def make(n):
        d = base[rng.integers(0, 24, n)].copy()  # many exact duplicates: ties across chunks
        d[rng.random(d.shape) < 0.02] ^= 0x11
        return d

    def rows64(d):
        r = np.zeros((len(d), 64), np.uint8)
        r[:, :61] = d
        return r

    q = make(600)
    sets = [make(n) for n in (1500, 1, 127, 0, 385, 2600)]
    dq = torch.from_numpy(rows64(q)).cuda()
    cat = torch.from_numpy(np.concatenate([rows64(t) for t in sets])).cuda()
    try:
        ctx.debug_set_match_chunks(0, chunks_per_set or 0)
        for ratio, thr in ((0.86, 10000), (1.3, 10000)):
            out, cnt = ctx.descriptor_match_sets_device(dq, cat, [len(t) for t in sets], thr, ratio)
            ctx.synchronize()
            out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
            for k, t in enumerate(sets):
                got = out[k][:int(cnt[k])].copy().view(ctx_match_dtype()).reshape(-1)
                exp = ref.descriptor_match(q, t, thr, ratio)
                assert np.array_equal(got, exp), (chunks_per_set, ratio, k, len(got), len(exp))
    finally:
        ctx.debug_set_match_chunks(0, 0)


@pytest.mark.parametrize("mode", [1, 0, 3])
def test_descriptor_match_sets(ctx, ref, mode):
    """One query set against several train sets in one launch (sizes around the tile sizes, an empty set, duplicates
    across sets): every set's match list equals the oracle's descriptor_match of that pair."""
    import torch
    rng = np.random.default_rng(77)
    base = rng.integers(0, 256, (40, 61), dtype=np.uint8)

    def make(n):
        d = base[rng.integers(0, 40, n)].copy()
        d[rng.random(d.shape) < 0.03] ^= 0x24
        return d

    q = make(700)
    sets = [make(n) for n in (129, 0, 1, 128, 1000, 37, 2051)]

    def rows64(d):
        r = np.zeros((len(d), 64), np.uint8)
        r[:, :61] = d
        return r

    dq = torch.from_numpy(rows64(q)).cuda()
    cat = torch.from_numpy(np.concatenate([rows64(t) for t in sets])).cuda()
    ctx.set_match_mode(mode)
    try:
        for ratio, thr in ((0.86, 10000), (1.2, 10000), (0.95, 9)):
            out, cnt = ctx.descriptor_match_sets_device(dq, cat, [len(t) for t in sets], thr, ratio)
            ctx.synchronize()
            out, cnt = out.cpu().numpy(), cnt.cpu().numpy()
            total = 0
            for k, t in enumerate(sets):
                got = out[k][:int(cnt[k])].copy().view(ctx_match_dtype()).reshape(-1)
                exp = ref.descriptor_match(q, t, thr, ratio)
                assert np.array_equal(got, exp), (mode, ratio, thr, k, len(got), len(exp))
                total += len(exp)
            assert total > 100
        out, cnt = ctx.descriptor_match_sets_device(dq[:0], cat, [len(t) for t in sets])
        assert int(cnt.sum().item()) == 0
    finally:
        ctx.set_match_mode(2)

@pytest.mark.parametrize("mode", [3, 1, 0])
@pytest.mark.parametrize("n_q", [700, 90, 5000])
def test_descriptor_match_sets_mutual(ctx, ref, mode, n_q):
    """Both directions of every (query set, train set) block from one pass (akz_descriptor_match_sets_mutual_device): the
    query direction as test_descriptor_match_sets, and for every train set its rows as queries against the query set --
    equal to the oracle's descriptor_match with the arguments exchanged.  Many exact duplicates (ties must resolve to the
    lowest index in BOTH directions), query sets shorter than / equal to / far beyond the seed rows, set sizes around
    the tile height, an empty set, thresholds that cut, several query blocks (concurrent updates of one train row)."""
    import torch
    rng = np.random.default_rng(1234 + n_q)
    base = rng.integers(0, 256, (48, 61), dtype=np.uint8)

    def make(n):
        d = base[rng.integers(0, 48, n)].copy()
        d[rng.random(d.shape) < 0.03] ^= 0x42
        return d

    def rows64(d):
        r = np.zeros((len(d), 64), np.uint8)
        r[:, :61] = d
        return r

    q = make(n_q)
    sets = [make(n) for n in (129, 0, 1, 128, 1000, 37, 2051, 640)]
    dq = torch.from_numpy(rows64(q)).cuda()
    cat = torch.from_numpy(np.concatenate([rows64(t) for t in sets])).cuda()
    rows = [len(t) for t in sets]
    ctx.set_match_mode(mode)
    try:
        for ratio, thr in ((0.86, 10000), (1.2, 10000), (0.95, 9), (0.86, 2 ** 63 - 1)):  # (the last: no threshold at all)
            out, cnt, cout, ccnt = ctx.descriptor_match_sets_mutual_device(dq, cat, rows, thr, ratio)
            ctx.synchronize()
            out, cnt, cout, ccnt = out.cpu().numpy(), cnt.cpu().numpy(), cout.cpu().numpy(), ccnt.cpu().numpy()
            off = total = 0
            for k, t in enumerate(sets):
                got = out[k][:int(cnt[k])].copy().view(ctx_match_dtype()).reshape(-1)
                assert np.array_equal(got, ref.descriptor_match(q, t, thr, ratio)), (mode, ratio, thr, k, "query direction")
                gotc = cout[off:off + int(ccnt[k])].copy().view(ctx_match_dtype()).reshape(-1)
                expc = ref.descriptor_match(t, q, thr, ratio)
                assert np.array_equal(gotc, expc), (mode, n_q, ratio, thr, k, len(gotc), len(expc), "opposite direction")
                off += len(t)
                total += len(expc)
            assert total > 50
        out, cnt, cout, ccnt = ctx.descriptor_match_sets_mutual_device(dq[:0], cat, rows)
        assert int(cnt.sum().item()) == 0 and int(ccnt.sum().item()) == 0
    finally:
        ctx.set_match_mode(2)



def ctx_match_dtype():
    import akaze_amd
    return akaze_amd.MATCH_DTYPE


@pytest.mark.parametrize("n1", [1, 2, 33, 129, 300])
def test_descriptor_match_unbounded_threshold_tiny_train_sets(mctx, ref, n1):
    """distance_threshold = usize::MAX-like values (feature_matching.rs:23-30 takes a usize) with train sets of one row, two
    rows and sizes that leave most of the last tile as padding: the second distance of a query then stays at the threshold
    (a huge number) or comes from a real row, never from a padding row -- lists equal the oracle's for every kernel."""
    ctx = mctx
    rng = np.random.default_rng(1000 + n1)
    d0 = rng.integers(0, 256, (700, 61), dtype=np.uint8)
    d1 = rng.integers(0, 256, (n1, 61), dtype=np.uint8)
    d1[0] = d0[5]
    if n1 > 1:
        d1[n1 - 1] = d0[5]          # an exact tie between the first and the last row: the lower row wins
        d1[n1 // 2] = d0[640] ^ 1
    for thr in (2 ** 63 - 1, 2 ** 31, 489, 244):
        for ratio in (0.86, 1.0):
            got = ctx.descriptor_match(d0, d1, thr, ratio)
            exp = ref.descriptor_match(d0, d1, thr, ratio)
            assert np.array_equal(got, exp), (n1, thr, ratio, len(got), len(exp))


def test_descriptor_match_full_width_rows(ctx, ref):
    """Descriptors of 62..64 bytes use the bytes the matrix-core kernel keeps its row counts in: the host entry point
    routes them to the popcount kernel."""
    rng = np.random.default_rng(5)
    for nb in (64, 62, 61, 21):
        d0 = rng.integers(0, 256, (200, nb), dtype=np.uint8)
        d1 = rng.integers(0, 256, (333, nb), dtype=np.uint8)
        d1[::3] = d0[:111]
        d1[::3, nb - 1] ^= 0x80
        got = ctx.descriptor_match(d0, d1, 10000, 0.86)
        exp = ref.descriptor_match(d0, d1, 10000, 0.86)
        assert len(exp) > 50 and np.array_equal(got, exp), nb


def test_descriptor_match_chunked_large(mctx, ref):
    """Train sets large enough to be split over several workgroups (chunk merge) with planted ties across
    chunk boundaries: result identical to the sequential scan of the oracle."""
    ctx = mctx
    rng = np.random.default_rng(21)
    d0 = rng.integers(0, 256, (700, 61), dtype=np.uint8)
    d1 = rng.integers(0, 256, (9001, 61), dtype=np.uint8)
    for i in range(0, 700, 2):
        j = (i * 13 + 5) % 9001
        d1[j] = d0[i]
        d1[j, i % 61] ^= 0x3
    # exact duplicates far apart (different chunks): the tie must resolve to the lower index and the ratio
    # test must reject them (second == min)
    d1[100] = d0[1]; d1[8000] = d0[1]
    d1[4097] = d0[3]; d1[4095] = d0[3]
    for ratio, thr in ((0.86, 10000), (0.7, 10000), (0.99, 60)):
        got = ctx.descriptor_match(d0, d1, thr, ratio)
        exp = ref.descriptor_match(d0, d1, thr, ratio)
        assert len(exp) > 100 and np.array_equal(got, exp)
    assert 1 not in got["index_0"] and 3 not in got["index_0"]


@pytest.mark.parametrize("n0,n1", [(2048, 300), (2049, 4097), (11264, 2500), (30001, 777)])
def test_descriptor_match_large_query_sets_one_launch_epilogue(mctx, ref, n0, n1):
    """Query sets of 2 048 rows and more take k_match_merge_compact: the merge of the chunk records, the ratio test and the
    ORDERED compaction in one launch of many workgroups (a look-back over the workgroups' counts, told apart from the counts
    of earlier calls by an epoch).  Sizes on both sides of the 256-row workgroups, accepted matches in some workgroups only
    (so that most counts are zero), exact duplicates across chunks, and the same call twice (the state words of the first
    call must read as "not yet published" in the second): identical to the oracle's sequential scan
    (feature_matching.rs:37-81), in every matcher kernel."""
    ctx = mctx
    rng = np.random.default_rng(n0 * 31 + n1)
    d0 = rng.integers(0, 256, (n0, 61), dtype=np.uint8)
    d1 = rng.integers(0, 256, (n1, 61), dtype=np.uint8)
    hot = np.concatenate([np.arange(0, min(n0, 300)), np.arange(n0 // 2, min(n0, n0 // 2 + 700)), np.arange(max(0, n0 - 5), n0)])
    for i in hot[::2]:   # near-duplicates: these queries pass the ratio test; the workgroups in between accept nothing
        j = int((i * 13 + 5) % n1)
        d1[j] = d0[i]
        d1[j, int(i % 61)] ^= 0x5
    if n1 > 600:
        d1[7] = d0[hot[1]]; d1[n1 - 3] = d0[hot[1]]   # an exact tie far apart: lowest index wins, the ratio test rejects it
    for ratio, thr in ((0.86, 10000), (0.95, 30)):
        exp = ref.descriptor_match(d0, d1, thr, ratio)
        got = ctx.descriptor_match(d0, d1, thr, ratio)
        again = ctx.descriptor_match(d0, d1, thr, ratio)
        assert np.array_equal(got, exp) and np.array_equal(again, exp), (ratio, thr, len(got), len(exp))
    assert len(ref.descriptor_match(d0, d1, 10000, 0.86)) > 50


def test_device_match_ignores_padding_bytes_in_both_kernels(ctx, ref):
    """akz_descriptor_match_device takes 64-byte rows of which an M-LDB descriptor uses 61: bytes 61..63 are padding and
    must not be compared by EITHER kernel (popcount and matrix-core), whatever they hold."""
    import torch
    rng = np.random.default_rng(77)
    d0 = rng.integers(0, 256, (300, 61), dtype=np.uint8)
    d1 = rng.integers(0, 256, (500, 61), dtype=np.uint8)
    d1[::4][:100] = d0[:100]
    d1[::4][:100, 7] ^= 0x11
    exp = ref.descriptor_match(d0, d1, 10000, 0.86)
    assert len(exp) > 50
    r0 = np.concatenate([d0, rng.integers(1, 256, (300, 3), dtype=np.uint8)], axis=1)  # garbage in the padding
    r1 = np.concatenate([d1, rng.integers(1, 256, (500, 3), dtype=np.uint8)], axis=1)
    for mode in (0, 1, 3):
        ctx.set_match_mode(mode)
        try:
            out, cnt = ctx.descriptor_match_device(torch.from_numpy(r0).cuda(), torch.from_numpy(r1).cuda(), 10000, 0.86)
            torch.cuda.synchronize()
            n = int(cnt.cpu()[0])
            got = out.cpu().numpy()[:n].reshape(-1).view(ctx_match_dtype())
            assert np.array_equal(got, exp), mode
        finally:
            ctx.set_match_mode(2)
