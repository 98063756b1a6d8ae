#!/usr/bin/env python3
"""How fast does a 2 x 3840x2160 u8 pair (16.6 MB, pinned) reach the device as one copy, and split over 2 / 4 streams?
(the upload is 0.65 ms of BASELINE configs[2]'s 3.2 ms extract call)"""
import time
import torch
n = 2 * 3840 * 2160
src = torch.empty(n, dtype=torch.uint8).pin_memory(); src.random_(0, 255)
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
streams = [torch.cuda.Stream() for _ in range(8)]
def run(parts, reps=200):
    cut = [n * i // parts for i in range(parts + 1)]
    for _ in range(10):
        for i in range(parts):
            with torch.cuda.stream(streams[i]): dst[cut[i]:cut[i + 1]].copy_(src[cut[i]:cut[i + 1]], non_blocking=True)
        torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        for i in range(parts):
            with torch.cuda.stream(streams[i]): dst[cut[i]:cut[i + 1]].copy_(src[cut[i]:cut[i + 1]], non_blocking=True)
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
for parts in (1, 2, 4, 8, 1, 2, 4):
    s = run(parts)
    print(f"{parts} stream(s): {s * 1e3:.3f} ms  {n / s / 1e9:.1f} GB/s", flush=True)
assert torch.equal(dst.cpu(), src)
