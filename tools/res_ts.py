"""k_octave_resident phase timestamps (AKZ_RES_DBG=2): one 1080p frame, a few calls."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
dev = torch.device("cuda", 0)
frames = torch.from_numpy(np.stack([A.synth_frame(1920, 1080, i) for i in range(1)])).to(dev)
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
for _ in range(4):
    ctx.extract_begin(frames[0:1], A.Config()).finish().close()
