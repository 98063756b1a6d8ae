#!/usr/bin/env python3
"""Frames in pinned host memory: the staged upload (akz_extract_begin_host_u8: one copy, then the chain) against handing the
pinned pointer to the device entry point (the first kernel reads the frames over the link itself).
    python tools/zero_copy_probe.py [W H N]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
W, H, N = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3840, 2160, 2)
frames = np.stack([A.synth_frame(W, H, i) for i in range(N)])
pinned = torch.from_numpy(frames).pin_memory()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream); ctx.warmup()
cfg = A.Config()
def staged(): return ctx.extract_begin_host(pinned, cfg).finish()
def direct():
    job = C.c_void_p()
    A._check(A.lib().akz_extract_begin_device_u8(ctx._h, C.c_void_p(pinned.data_ptr()), W, H, N, C.byref(cfg), A.AKZ_KEEP_ALL_PLANES, C.byref(job)))
    return A.Job(ctx, job, pinned).finish()
def sig(r): return [(r.keypoints(i).tobytes(), r.descriptors(i).tobytes()) for i in range(N)]
a = staged(); b = direct(); print("same results:", sig(a) == sig(b), a.counts(0)); a.close(); b.close()
for rnd in range(3):
    for name, fn in (("staged", staged), ("direct", direct)):
        for _ in range(5): fn().close()
        t = time.perf_counter()
        for _ in range(40): fn().close()
        print(f"{N}x{W}x{H} {name}: {(time.perf_counter() - t) / 40 * 1e3:.3f} ms per call", flush=True)
