#!/usr/bin/env python3
"""Soak of the keypoint selection on the device (k_select): random jobs -- sizes up to 2.5 Mpx, batches of 1-8, synthetic / noise /
blended / flat frames, thresholds over four decades, pyramid depths -- each through the device's selection and through the host's
(from the device's neighbour lists), synchronous and pipelined; counts, keypoints and descriptors must be identical.
    python tools/select_soak.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
c = A.Context(0, st.cuda_stream)
t0 = time.time(); jobs = images = on_dev = fell = kps = 0; most_looks = 0
def sig(r, n): return [(r.counts(i), r.keypoints(i).tobytes(), r.descriptors(i).tobytes()) for i in range(n)]
while time.time() - t0 < budget:
    big = rng.integers(0, 6) == 0
    w, h = (int(rng.integers(900, 2000)), int(rng.integers(700, 1300))) if big else (int(rng.integers(96, 800)), int(rng.integers(96, 600)))
    n = int(rng.integers(1, 3 if big else 9))
    frames = []
    for i in range(n):
        kind = int(rng.integers(0, 5))
        syn = A.synth_frame(w, h, int(rng.integers(0, 100000)))
        noi = rng.integers(0, 256, (h, w), dtype=np.uint8)
        frames.append(syn if kind <= 1 else noi if kind == 2 else ((syn.astype(np.uint16) + noi) // 2).astype(np.uint8) if kind == 3
                      else np.full((h, w), int(rng.integers(0, 256)), np.uint8))
    kw = {}
    if rng.integers(0, 3) == 0: kw["detector_threshold"] = float(10.0 ** rng.uniform(-6.5, -2.5))
    if rng.integers(0, 4) == 0 and min(w, h) >= 200: kw.update(num_sublevels=int(rng.integers(2, 6)), max_octave_evolution=int(rng.integers(2, 5)))
    if rng.integers(0, 5) == 0: kw["descriptor_channels"] = int(rng.integers(1, 4))
    cfg = A.Config(**kw)
    dev = torch.from_numpy(np.stack(frames)).cuda()
    c.debug_set_select(1)
    r1 = c.extract_features(dev, cfg, keep_all_planes=False); want = sig(r1, n); r1.close()
    c.debug_set_select(2)
    r2 = c.extract_features(dev, cfg, keep_all_planes=False); info = c.debug_select_info(); got = sig(r2, n); r2.close()
    if rng.integers(0, 3) == 0:  # ... and two jobs in flight
        ja, jb = c.extract_begin(dev, cfg, keep_all_planes=False), c.extract_begin(dev, cfg, keep_all_planes=False)
        for j in (ja, jb):
            r = j.finish(); assert sig(r, n) == want, ("pipelined", w, h, n, kw); r.close()
    if got != want:
        print("MISMATCH", w, h, n, kw, info[:4], [g[0] for g in got], [x[0] for x in want]); sys.exit(1)
    jobs += 1; images += n; on_dev += info[0] == 2; fell += info[0] != 2; kps += sum(g[0][1] for g in got); most_looks = max(most_looks, info[1])
    if jobs % 50 == 0: print(f"{time.time() - t0:6.0f} s  {jobs} jobs, {images} images, {kps} keypoints, {on_dev} on the device, {fell} back on the host, most looks {most_looks}", flush=True)
print(f"select soak: {jobs} jobs, {images} images, {kps} keypoints identical; {on_dev} jobs selected on the device, {fell} went back to the host; most looks of a thread {most_looks}")
