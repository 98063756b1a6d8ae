#!/usr/bin/env python3
"""Known-byte-count kernels for calibrating rocprofv3 FETCH_SIZE / WRITE_SIZE on this box: a 512 MiB float32 copy
(torch, 16 B/lane), the FED kernel on one 3840x2160 plane, and the detector column march (8 B/lane loads, 8 B/lane
streaming stores) on a 32 x 1920x1080 level whose read and write byte counts are known exactly."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
a = torch.rand(128 * 1024 * 1024, device="cuda")   # 512 MiB
b = torch.empty_like(a)
for _ in range(3):
    b.copy_(a)
torch.cuda.synchronize()
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
lt = torch.rand((2160, 3840), device="cuda"); lf = torch.rand((2160, 3840), device="cuda")
ctx.fed_steps(lt, lf, np.full(40, 0.2))            # 5 launches x 8 steps
ctx.set_fed_mode(0)
ctx.fed_steps(lt, lf, np.full(4, 0.2))             # 4 single-step launches
ctx.set_fed_mode(2)
ls = torch.rand((32, 1080, 1920), device="cuda")
ctx.set_detector_mode(5)
for _ in range(3):
    ctx.detector_response(ls, 3)                   # k_detector_march<3, false, true>: 265.4 MB plane read once (+ halos), 6 planes written
torch.cuda.synchronize()
