"""CPU tests of the oracle: the reference's own known-answer literals, the filter border rule,
FED step-size vectors and the committed golden fixtures.  No GPU, no product code."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_gaussian_kernel_reference_kat(ref):
    # akaze/src/types/image.rs:486-502 (gaussian_kernel_correct), tolerance as in the reference
    known = np.array([0.10628852, 0.14032133, 0.16577007, 0.17524014, 0.16577007, 0.14032133, 0.10628852],
                     np.float32)
    got = ref.gaussian_kernel(3.0, 7)
    assert np.all(np.abs(got - known) < 1e-4)
    assert abs(float(got.sum()) - 1.0) < 1e-6


def test_scharr_kernel_reference_literals(ref):
    # akaze/src/ops/derivatives.rs:11-28 (literals; the reference's relative_eq! result is discarded)
    m, o = ref.scharr_kernels(1)
    assert np.array_equal(m, np.array([0.09375, 0.3125, 0.09375], np.float32))
    assert np.array_equal(o, np.array([-1.0, 0.0, 1.0], np.float32))
    # larger scales: 3 non-zero taps at -s, 0, +s (derivatives.rs:74-101)
    for s, (n, wn) in {2: (0.046875, 0.15625), 3: (0.03125, 0.104166664), 4: (0.0234375, 0.078125)}.items():
        m, o = ref.scharr_kernels(s)
        assert len(m) == 2 * s + 1
        assert m[0] == np.float32(n) and m[-1] == np.float32(n) and m[s] == np.float32(wn)
        assert np.count_nonzero(m) == 3 and np.count_nonzero(o) == 2 and o[0] == -1 and o[-1] == 1


@pytest.mark.parametrize("shape,klen", [((11, 13), 3), ((12, 17), 5), ((19, 20), 9), ((40, 33), 7)])
def test_filter_border_rule(ref, shape, klen):
    """image.rs:239-332: out(x,y) = valid-interior result at coordinates clamped to [hw, dim-1-hw],
    both passes clamp in both dimensions — checked against a direct numpy statement."""
    rng = np.random.default_rng(klen)
    img = rng.standard_normal(shape).astype(np.float32)
    kern = rng.standard_normal(klen).astype(np.float32)
    hw = klen // 2
    h, w = shape

    def direct(horizontal):
        out = np.zeros(shape, np.float32)
        for y in range(h):
            for x in range(w):
                cx, cy = min(max(x, hw), w - 1 - hw), min(max(y, hw), h - 1 - hw)
                acc = np.float32(0)
                for i in range(klen):
                    v = img[cy, cx + i - hw] if horizontal else img[cy + i - hw, cx]
                    acc = np.float32(acc + np.float32(kern[i] * v))
                out[y, x] = acc
        return out

    assert np.array_equal(ref.horizontal_filter(img, kern), direct(True))
    assert np.array_equal(ref.vertical_filter(img, kern), direct(False))


def test_half_size_order(ref):
    rng = np.random.default_rng(1)
    img = rng.standard_normal((9, 11)).astype(np.float32)
    out = ref.half_size(img)
    assert out.shape == (4, 5)
    a = img[0:8:2, 0:10:2]; b = img[1:9:2, 0:10:2]; c = img[0:8:2, 1:11:2]; d = img[1:9:2, 1:11:2]
    exp = ((((np.float32(0) + a) + b) + c) + d) / np.float32(4)
    assert np.array_equal(out, exp.astype(np.float32))


def test_fed_tau_kats(ref):
    # SURVEY.md Appendix B, derived by hand-simulating akaze/src/ops/fed_tau.rs
    t1 = ref.fed_tau(0.53019335983756166)
    assert np.allclose(t1, [0.069726728405419838, 0.10842214335785509, 0.35204448807428662], rtol=1e-14, atol=0)
    t3 = ref.fed_tau(1.0603867196751229)
    assert np.allclose(t3, [0.10603867196751227, 0.67986420186701202, 0.082001656472159318, 0.19248218936843872],
                       rtol=1e-14, atol=0)
    t8 = ref.fed_tau(5.9984531212995087)
    assert len(t8) == 8 and np.allclose(t8[:3], [0.19623365730888334, 3.7012260958150769, 0.12604081556156974],
                                        rtol=1e-14)
    assert abs(t8.sum() - 5.9984531212995087) < 1e-12
    with pytest.raises(ValueError):  # n == 1: the reference underflows a usize (fed_tau.rs:95)
        ref.fed_tau(0.1)


def test_level_table_default_config(ref):
    # SURVEY.md Appendix B level table for Config::default() on a 1080p frame
    img = np.zeros((1080, 1920), np.float32)
    img[::7, ::5] = 1.0
    # only the plan is of interest: use a tiny image through the same planner instead
    r = ref.extract(np.zeros((135, 240), np.uint8) + 7)
    assert r.num_levels == 8  # 240x135 admits two octaves (120x67 >= 80x40)
    ns = [len(r.level_info(i)["tau"]) for i in range(8)]
    assert ns == [0, 3, 3, 4, 4, 5, 6, 7]
    assert [r.level_info(i)["w"] for i in range(8)] == [240] * 4 + [120] * 4


def test_fed_step_border_cases(ref):
    """nonlinear_diffusion.rs:30-143 against an independent numpy statement of the 9 cases."""
    rng = np.random.default_rng(5)
    h, w = 7, 9
    L = rng.random((h, w)).astype(np.float32)
    c = rng.random((h, w)).astype(np.float32)
    tau = 3.7012260958150769
    got, step = ref.fed_step(L, c, tau)
    f = np.float32
    ht = f(0.5) * f(tau)
    exp = np.zeros_like(L)
    for y in range(h):
        for x in range(w):
            xpos = f(f(c[y, x] + c[y, x + 1]) * f(L[y, x + 1] - L[y, x])) if x + 1 < w else None
            xneg = f(f(c[y, x - 1] + c[y, x]) * f(L[y, x] - L[y, x - 1])) if x > 0 else None
            ypos = f(f(c[y, x] + c[y + 1, x]) * f(L[y + 1, x] - L[y, x])) if y + 1 < h else None
            yneg = f(f(c[y - 1, x] + c[y, x]) * f(L[y, x] - L[y - 1, x])) if y > 0 else None
            if ypos is None:  # last row: y_pos is taken towards y-1
                ypos = f(f(c[y, x] + c[y - 1, x]) * f(L[y - 1, x] - L[y, x]))
                yneg = None
            t = f(xpos - xneg) if (xpos is not None and xneg is not None) else (xpos if xpos is not None else f(-xneg))
            t = f(t + ypos)
            if yneg is not None:
                t = f(t - yneg)
            exp[y, x] = f(ht * t)
    assert np.array_equal(step, exp)
    assert np.array_equal(got, (L + exp).astype(np.float32))


def test_descriptor_match_semantics(ref):
    """feature_matching.rs:37-81 == exact brute force with lowest-index ties, ratio^2 and threshold."""
    rng = np.random.default_rng(3)
    d0 = rng.integers(0, 256, (40, 61), dtype=np.uint8)
    d1 = rng.integers(0, 256, (55, 61), dtype=np.uint8)
    d1[7] = d0[3]; d1[20] = d0[3]          # exact duplicates: tie -> first index, second == min -> rejected
    d1[9] = d0[5]; d1[9, 0] ^= 1            # near duplicate -> accepted
    got = ref.descriptor_match(d0, d1, 10000, 0.86)
    pop = np.unpackbits(d0[:, None, :] ^ d1[None, :, :], axis=2).sum(axis=2)
    exp = []
    for i in range(len(d0)):
        order = np.argsort(pop[i], kind="stable")
        mn, sc = pop[i][order[0]], pop[i][order[1]]
        if float(mn) < float(sc) * 0.86 ** 2 and mn < 10000:
            exp.append((i, order[0], float(mn)))
    assert [(int(m["index_0"]), int(m["index_1"]), float(m["distance"])) for m in got] == exp
    assert 5 in got["index_0"] and 3 not in got["index_0"]
    # edge cases: empty train set -> nothing; single train descriptor -> second stays at the threshold
    assert len(ref.descriptor_match(d0, np.zeros((0, 61), np.uint8))) == 0
    assert len(ref.descriptor_match(d0[:4], d1[:1])) == 4
    assert len(ref.descriptor_match(d0[:4], d1[:1], distance_threshold=10)) == 0


def test_golden_fixtures(ref):
    """Committed vectors (tests/golden/make_golden.py): guards the oracle against silent drift."""
    import hashlib
    path = os.path.join(GOLDEN, "synthetic_small.npz")
    g = np.load(path)
    for name in ("a", "b"):
        frame = g[f"{name}_frame"]
        r = ref.extract(frame)
        assert r.num_keypoints == int(g[f"{name}_num_keypoints"])
        assert np.array_equal(r.keypoints().view(np.uint8), g[f"{name}_keypoints"])
        assert np.array_equal(r.descriptors(), g[f"{name}_descriptors"])
        assert r.contrast == float(g[f"{name}_contrast"])
        sums = []
        for lvl in range(r.num_levels):
            for pl in ("Lt", "Lsmooth", "Lflow", "Ldet"):
                sums.append(hashlib.sha256(r.plane(lvl, pl).tobytes()).hexdigest()[:16])
        assert sums == list(g[f"{name}_plane_sha"])
    m = ref.descriptor_match(g["a_descriptors"], g["b_descriptors"], 10000, 0.86)
    assert np.array_equal(m.view(np.uint8), g["matches_ab"])
