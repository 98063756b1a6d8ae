"""Exploratory check: streaming detector kernels (mode 1) against the tiled ones (mode 0) on the GPU."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import torch
import akaze_amd as A

PL = ["Lx", "Ly", "Lxx", "Lyy", "Lxy", "Ldet", "Lt", "Lflow"]
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
bad = 0
for (w, h, n) in [(320, 240, 1), (517, 389, 2), (640, 480, 3), (130, 96, 2), (1920, 1080, 2), (1001, 777, 1)]:
    frames = np.stack([A.synth_frame(w, h, i) for i in range(n)])
    d = torch.from_numpy(frames).cuda()
    res = {}
    for mode in (0, 1):
        ctx.set_detector_mode(mode)
        ctx.set_prep_mode(mode)
        res[mode] = ctx.extract_features(d)
    for i in range(n):
        nl = res[0].counts(i)[0]
        for lvl in range(nl):
            for pl in PL:
                a, b = res[0].plane(lvl, pl, i), res[1].plane(lvl, pl, i)
                if not np.array_equal(a, b):
                    nb = np.argwhere(a != b)
                    print(f"MISMATCH {w}x{h} img {i} level {lvl} {pl}: {len(nb)} px, first {nb[0]}, rows {np.unique(nb[:,0])[:8]} cols {np.unique(nb[:,1])[:8]}")
                    bad += 1
        k0, k1 = res[0].keypoints(i), res[1].keypoints(i)
        same = k0.tobytes() == k1.tobytes() and np.array_equal(res[0].descriptors(i), res[1].descriptors(i))
        print(f"{w}x{h} img {i}: {len(k0)} / {len(k1)} keypoints, identical: {same}")
        bad += not same
for sigma in (1, 2, 3, 4):
    for (w, h, n) in [(130, 96, 3), (517, 200, 1)]:
        rng = np.random.default_rng(sigma)
        ls = torch.from_numpy(rng.random((n, h, w), dtype=np.float32)).cuda()
        out = {}
        for mode in (0, 1):
            ctx.set_detector_mode(mode)
            out[mode] = ctx.detector_response(ls, sigma)
        for k in out[0]:
            if not torch.equal(out[0][k], out[1][k]):
                nb = (out[0][k] != out[1][k]).nonzero()
                print(f"OP MISMATCH sigma {sigma} {w}x{h} {k}: {len(nb)} first {nb[0].tolist()}")
                bad += 1
print("bad =", bad)
