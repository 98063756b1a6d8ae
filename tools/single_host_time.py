import os, sys, time
sys.path.insert(0, "akaze-rust_amd/python")
import numpy as np, torch
import akaze_amd as A
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream)
one = torch.from_numpy(A.synth_frame(1920, 1080, 0)[None]).cuda()
for lanes in (1, 2, 4):
    ctx.set_lanes(lanes)
    for _ in range(10):
        ctx.extract_begin(one).finish().close()
    tb = tf = 0.0; reps = 200
    pending = []
    t0 = time.perf_counter()
    for _ in range(reps):
        a = time.perf_counter(); pending.append(ctx.extract_begin(one)); tb += time.perf_counter() - a
        if len(pending) > lanes:
            a = time.perf_counter(); pending.pop(0).finish().close(); tf += time.perf_counter() - a
    while pending:
        a = time.perf_counter(); pending.pop(0).finish().close(); tf += time.perf_counter() - a
    tot = time.perf_counter() - t0
    print(f"lanes {lanes}: {tot/reps*1e3:.3f} ms/frame  host in begin {tb/reps*1e3:.3f}  in finish {tf/reps*1e3:.3f}")
