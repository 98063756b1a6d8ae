#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU, the matcher half alone: N 3840x2160 5x5 frames extracted once, then the gather +
akz_match_all_pairs step timed `reps` times.  python tools/c5_leg.py [frames] [reps] [chunks per set]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
n_fr = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
set_chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # chunks per set of the multi-set launches (0: automatic)
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
cfg = A.Config(num_sublevels=5, max_octave_evolution=5)
ress = []
for k in range(0, n_fr, 4):
    fr = torch.from_numpy(np.stack([A.synth_frame(3840, 2160, i) for i in range(k, min(n_fr, k + 4))])).cuda()
    ress.append(ctx.extract_features(fr, cfg, keep_all_planes=False, host_descriptors=False))
rows = sum(r.counts(i)[1] for r in ress for i in range(r.num_images))
comm = A.Comm(0, A.comm_unique_id(), 0, 1)
if set_chunks: ctx.debug_set_match_chunks(0, set_chunks)
for mode in (2, 1):
    ctx.set_match_mode(mode)
    best = 1e9
    for it in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g = comm.gather_begin(ress, rows + 64)
        pr = g.match_all_pairs(ctx)
        pr.count(0, 1)
        dt = (time.perf_counter() - t0) * 1e3
        tot = pr.total_matches()
        pairs_n = sum(pr.image_rows(q)[0] * pr.image_rows(j)[0] for q in range(pr.n_images) for j in range(pr.n_images) if q != j)
        pr.free(); g.free()
        if it: best = min(best, dt)
    print(f"mode {mode}: {n_fr} frames, {rows} rows, {tot} matches in {n_fr * (n_fr - 1)} ordered pairs: gather + all-pairs {best:.2f} ms "
          f"({pairs_n / best / 1e9:.2f} T ordered pairs/s)")
