#!/usr/bin/env python3
"""A stream of lone 1920x1080 frames on one context dealt to 1 .. 6 lanes (frames in flight = lanes), three rounds:
how the per-frame time moves with the lane count and from round to round.  python tools/lanes1080.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
fr = torch.from_numpy(A.synth_frame(1920, 1080, 3)[None]).cuda()
torch.cuda.synchronize()
def stream(lanes, reps=200):
    ctx.set_lanes(lanes)
    for _ in range(2 * lanes):
        ctx.extract_begin(fr, keep_all_planes=True).finish().close()
    pend = []
    t0 = time.perf_counter()
    for _ in range(reps):
        pend.append(ctx.extract_begin(fr, keep_all_planes=True))
        if len(pend) >= max(lanes, 1):
            pend.pop(0).finish().close()
    while pend:
        pend.pop(0).finish().close()
    return (time.perf_counter() - t0) / reps * 1e3
for rnd in range(3):
    print("round", rnd, "  ".join(f"{l} lanes {stream(l):.3f} ms" for l in (1, 2, 3, 4, 5, 6)), ctx.debug_stream_placement() if hasattr(ctx, "debug_stream_placement") else "")
