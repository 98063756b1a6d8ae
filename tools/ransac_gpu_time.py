#!/usr/bin/env python3
"""match_features on a 4K pair's features: wall time of the call and of its parts (descriptor_match alone; the host-only
remove_outliers for comparison).  python tools/ransac_gpu_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
f0, f1 = A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))
r0, r1 = ctx.extract_features(f0, keep_all_planes=False), ctx.extract_features(f1, keep_all_planes=False)
k0, d0, k1, d1 = r0.keypoints(), r0.descriptors(), r1.keypoints(), r1.descriptors()
def t(fn, reps=30):
    fn(); fn()
    a = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - a) / reps * 1e3
raw = ctx.descriptor_match(d0, d1, 10000, 0.86)
print(f"{len(raw)} matches; descriptor_match {t(lambda: ctx.descriptor_match(d0, d1, 10000, 0.86)):.3f} ms; "
      f"match_features (trials on the device) {t(lambda: A.match_features(k0, d0, k1, d1, 0.86, 1000, 3.0, ctx=ctx)):.3f} ms; "
      f"remove_outliers alone (trials on host threads) {t(lambda: A.remove_outliers(k0, k1, raw, 1000, 0.05, 3.0)):.3f} ms")
