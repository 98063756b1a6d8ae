#!/usr/bin/env python3
"""The pair matcher at the 4K-pair size under forced chunk counts (akz_debug_set_match_chunks)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "akaze-rust_amd", "python"))
import torch
import akaze_amd as A
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream)
g = torch.Generator(device="cuda").manual_seed(7)
for n in (11264, 5000, 30000):
    da = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g); da[:, 61:] = 0
    db = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g); db[:, 61:] = 0
    for ch in (0, 2, 3, 4, 5, 6, 8, 11, 16, 22, 32):
        ctx.debug_set_match_chunks(ch, 0)
        for _ in range(5): ctx.descriptor_match_device(da, db)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ctx.descriptor_match_device(da, db)
        e1.record(); torch.cuda.synchronize()
        print(f"n={n} chunks={ch or 'auto':>4}: {e0.elapsed_time(e1) / 30 * 1e3:7.1f} us")
