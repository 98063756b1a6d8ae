#!/usr/bin/env python3
"""The FP4 matrix-core matcher alone on one large pair (8 x the rows of a 3840x2160 frame each side), for counter runs:
rocprofv3 --pmc ... -- python3 tools/match_big.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import torch
import akaze_amd as A
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
f0, f1 = A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))
def rows(fr):
    res = ctx.extract_features(fr, keep_all_planes=False)
    n = sum(res.counts(i)[1] for i in range(res.num_images))
    t = torch.empty((n, 64), dtype=torch.uint8, device="cuda"); res.copy_device_descriptors(t); return t
big0, big1 = torch.cat([rows(f0)] * 8), torch.cat([rows(f1)] * 8)
ctx.set_match_mode(3)
ctx.descriptor_match_device(big0, big1); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): out, cnt = ctx.descriptor_match_device(big0, big1)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"n0={big0.shape[0]} n1={big1.shape[0]} {ms*1e3:.1f} us  {big0.shape[0]*big1.shape[0]/ms/1e9:.3f} T pairs/s")
