"""A randomised soak of the matrix-core matcher against the oracle (feature_matching.rs:23-94): random set sizes around
the tile / query-block / chunk boundaries, duplicates and near-duplicates (ties in both directions), random thresholds
and ratios, one-direction pair calls, multi-set launches and both-direction launches.  A short run is a GPU test; more
rounds: AKZ_MATCH_SOAK=400 python -m pytest tests/test_gpu_match_soak.py -m gpu."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sets(rng, n, base, flip):
    d = base[rng.integers(0, len(base), n)].copy()
    d[rng.random(d.shape) < flip] ^= rng.integers(1, 256, dtype=np.uint8)
    return d


def _rows64(d):
    r = np.zeros((len(d), 64), np.uint8)
    r[:, :61] = d
    return r


def test_matcher_random_soak(ctx, amd, ref):
    import torch
    rounds = int(os.environ.get("AKZ_MATCH_SOAK", "24"))
    rng = np.random.default_rng(int(os.environ.get("AKZ_MATCH_SOAK_SEED", "20260704")))
    sizes = [1, 2, 31, 32, 33, 127, 128, 129, 255, 257, 511, 512, 513, 1000, 1024, 1500, 2047, 2049, 3000, 5000]
    dt = amd.MATCH_DTYPE
    for it in range(rounds):
        base = rng.integers(0, 256, (int(rng.integers(8, 400)), 61), dtype=np.uint8)
        flip = float(rng.choice([0.0, 0.01, 0.05, 0.3]))
        n_q = int(rng.choice(sizes))
        q = _sets(rng, n_q, base, flip)
        n_sets = int(rng.integers(1, 7))
        sets = [_sets(rng, int(rng.choice([0] + sizes)), base, flip) for _ in range(n_sets)]
        thr = int(rng.choice([3, 40, 244, 489, 10000, 2 ** 63 - 1]))
        ratio = float(rng.choice([0.5, 0.86, 1.0, 1.3]))
        dq = torch.from_numpy(_rows64(q)).cuda()
        cat = torch.from_numpy(np.concatenate([_rows64(t) for t in sets] + [np.zeros((0, 64), np.uint8)])).cuda()
        rows = [len(t) for t in sets]
        exp_f = [ref.descriptor_match(q, t, thr, ratio) for t in sets]
        exp_r = [ref.descriptor_match(t, q, thr, ratio) for t in sets]
        # both directions from one launch
        out, cnt, cout, ccnt = ctx.descriptor_match_sets_mutual_device(dq, cat, rows, thr, ratio)
        ctx.synchronize()
        out, cnt, cout, ccnt = out.cpu().numpy(), cnt.cpu().numpy(), cout.cpu().numpy(), ccnt.cpu().numpy()
        off = 0
        for k in range(n_sets):
            got = out[k][:int(cnt[k])].copy().view(dt).reshape(-1)
            assert np.array_equal(got, exp_f[k]), ("mutual, query direction", it, n_q, rows, k, thr, ratio)
            gotc = cout[off:off + int(ccnt[k])].copy().view(dt).reshape(-1)
            assert np.array_equal(gotc, exp_r[k]), ("mutual, opposite direction", it, n_q, rows, k, thr, ratio)
            off += rows[k]
        # the one-direction multi-set launch and a pair call
        out1, cnt1 = ctx.descriptor_match_sets_device(dq, cat, rows, thr, ratio)
        ctx.synchronize()
        out1, cnt1 = out1.cpu().numpy(), cnt1.cpu().numpy()
        for k in range(n_sets):
            got = out1[k][:int(cnt1[k])].copy().view(dt).reshape(-1)
            assert np.array_equal(got, exp_f[k]), ("sets", it, n_q, rows, k, thr, ratio)
        k = int(rng.integers(0, n_sets))
        assert np.array_equal(ctx.descriptor_match(q, sets[k], thr, ratio), exp_f[k]), ("pair", it, n_q, rows[k], thr, ratio)
