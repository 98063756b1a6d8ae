#!/usr/bin/env python3
"""Six synchronous extract calls on a 2 x 3840x2160 batch, for a kernel trace of BASELINE configs[2]'s extraction half:
rocprofv3 --kernel-trace -- python3 tools/pair_trace.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
pair = np.stack([A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))])
d = torch.from_numpy(pair).cuda()
for _ in range(6):
    ctx.extract_begin(d).finish().close()
    time.sleep(0.01)
