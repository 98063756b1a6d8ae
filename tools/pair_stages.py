#!/usr/bin/env python3
"""BASELINE configs[2]'s extraction half alone: one synchronous extract call on a 2 x 3840x2160 batch from pinned host
memory (upload, scale space, keypoints, descriptors to the host), wall time and the library's stage clocks.
python tools/pair_stages.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
pair = np.stack([A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))])
h = torch.from_numpy(pair).pin_memory()
d = h.cuda()
def run(fn, reps=20):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
print(f"host input, descriptors to host, all planes : {run(lambda: ctx.extract_begin_host(h).finish().close()):.3f} ms")
print(f"device input, descriptors to host, all planes: {run(lambda: ctx.extract_begin(d).finish().close()):.3f} ms")
print(f"device input, descriptors stay on the device : {run(lambda: ctx.extract_begin(d, host_descriptors=False).finish().close()):.3f} ms")
print(f"device input, lean planes                   : {run(lambda: ctx.extract_begin(d, keep_all_planes=False).finish().close()):.3f} ms")
ctx.set_profiling(1)
for _ in range(10): ctx.extract_begin(d).finish().close()
p = ctx.get_profile(reset=True)
print("stage clocks (profiling on: every stage on one stream, nothing overlapped):", {k: (round(v / 10, 3) if isinstance(v, float) else v) for k, v in p.items()})
