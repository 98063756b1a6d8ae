#!/usr/bin/env python3
"""Experiment: C contexts (own stream + host thread each), every one running the begin/finish pipeline over
its share of F frames per step.  Prints Mpix/s for C = 1, 2, 3."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
W, H = 1920, 1080
def run(C, F, steps=6, warm=2):
    frames = torch.from_numpy(np.stack([A.synth_frame(W, H, i) for i in range(F)])).cuda()
    torch.cuda.synchronize()
    ctxs = [A.Context(0, torch.cuda.Stream().cuda_stream) for _ in range(C)]
    cut = [(F * i) // C for i in range(C + 1)]
    parts = [frames[cut[i]:cut[i + 1]] for i in range(C)]
    cfg = A.Config()
    def worker(c, k):
        prev = None
        for _ in range(k):
            job = ctxs[c].extract_begin(parts[c], cfg)
            if prev is not None: prev.finish().close()
            prev = job
        prev.finish().close()
    def go(k):
        th = [threading.Thread(target=worker, args=(c, k)) for c in range(C)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
    go(warm)
    t0 = time.perf_counter(); go(steps); dt = time.perf_counter() - t0
    print(f"contexts={C} frames/step={F}: {W*H*F*steps/dt/1e6:8.1f} Mpix/s  {dt/steps*1e3:6.2f} ms/step", flush=True)
    for c in ctxs: c.close()
for C, F in ((1, 32), (2, 32), (2, 64), (3, 48), (4, 64), (1, 64)):
    run(C, F)
