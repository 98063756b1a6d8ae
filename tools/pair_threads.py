#!/usr/bin/env python3
"""One synchronous extract call on a 2 x 3840x2160 batch under different host pool sizes (akz_ctx_set_host_threads), wall
time and the stage clocks: what waking the pool costs a small job.  python tools/pair_threads.py"""
import os, sys, time
sys.path.insert(0, "akaze-rust_amd/python")
import numpy as np, torch
import akaze_amd as A
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
pair = np.stack([A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))])
d = torch.from_numpy(pair).cuda()
def run(fn, reps=20):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
for th in (0, 1, 2, 4, 8):
    ctx.set_host_threads(th)
    print(f"host threads {th or 'auto'}: {run(lambda: ctx.extract_begin(d).finish().close()):.3f} ms per 2 x 4K call")
    ctx.set_profiling(1)
    for _ in range(10): ctx.extract_begin(d).finish().close()
    p = ctx.get_profile(reset=True)
    print("   ", {k: round(v / 10, 3) for k, v in p.items() if isinstance(v, float) and k in ("nms", "host_kp", "orient", "mldb", "total")})
    ctx.set_profiling(0)
