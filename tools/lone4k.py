#!/usr/bin/env python3
"""A stream of lone 3840x2160 frames on one context: begun 1 / 2 ahead, finish half on the caller's thread or the
context's own, and dealt to 4 lanes.  python tools/lone4k.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
fr = torch.from_numpy(A.synth_frame(3840, 2160, 3)[None]).cuda()
torch.cuda.synchronize()
def stream(depth, reps=40, ready=True):
    pend = []
    t0 = time.perf_counter()
    for _ in range(reps):
        pend.append(ctx.extract_begin(fr, keep_all_planes=True, input_ready=ready))
        if len(pend) > depth:
            pend.pop(0).finish().close()
    while pend:
        pend.pop(0).finish().close()
    return (time.perf_counter() - t0) / reps * 1e3
for name, lanes, eager, depth in (("plain, 1 ahead", 1, 0, 1), ("plain, 2 ahead", 1, 0, 2), ("own thread, 2 ahead", 1, 1, 2), ("4 lanes", 4, 0, 3)):
    ctx.set_lanes(lanes)
    ctx.set_eager_finish(bool(eager))
    stream(depth, 8)
    ms = min(stream(depth), stream(depth))
    print(f"{name:22s} {ms:.3f} ms per 4K frame ({3840 * 2160 / ms / 1e3:.0f} Mpix/s)")
