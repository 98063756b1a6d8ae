"""Kernel-level view of one descriptor_match_device call at the C3 size (11264 x 11264) and of the 16-set launch: run
under `rocprofv3 --kernel-trace --stats`."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "akaze-rust_amd", "python"))
import torch
import akaze_amd as A
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream)
g = torch.Generator(device="cuda").manual_seed(7)
n = 11264
da = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g); da[:, 61:] = 0
db = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g); db[:, 61:] = 0
dt = torch.randint(0, 256, (n * 16, 64), dtype=torch.uint8, device="cuda", generator=g); dt[:, 61:] = 0
for _ in range(20):
    ctx.descriptor_match_device(da, db)
torch.cuda.synchronize()
for _ in range(10):
    ctx.descriptor_match_sets_device(da, dt, [n] * 16)
torch.cuda.synchronize()
