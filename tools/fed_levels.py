"""FED launch timing by level shape and step count, for the kernel variants (op entry point, 32-frame batches).
The op entry point works in place for its caller and therefore copies the plane aside first; that device copy is
timed separately and subtracted.  Run through gpurun:  python tools/fed_levels.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np
import torch
import akaze_amd as A

main = torch.cuda.Stream()
with torch.cuda.stream(main):
    ctx = A.Context(0, main.cuda_stream)
    n = 32
    print(f"{'level':>12s} {'steps':>5s} " + " ".join(f"{m:>12s}" for m in ("own us", "fused us", "own TB/s", "copy us")))
    for (w, h) in ((1920, 1080), (960, 540), (480, 270), (240, 135)):
        lt = torch.rand((n, h, w), device="cuda", dtype=torch.float32)
        lf = torch.rand((n, h, w), device="cuda", dtype=torch.float32)
        scratch = torch.empty_like(lt)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        scratch.copy_(lt)
        e0.record(main)
        for _ in range(5):
            scratch.copy_(lt)
        e1.record(main)
        e1.synchronize()
        copy_us = e0.elapsed_time(e1) / 5 * 1e3
        for steps in (1, 2, 3, 4, 5, 6, 8):
            taus = np.full(steps, 0.2)
            row = []
            for mode in (2, 1):
                ctx.set_fed_mode(mode)
                for _ in range(2):
                    ctx.fed_steps(lt, lf, taus)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 5
                e0.record(main)
                for _ in range(reps):
                    ctx.fed_steps(lt, lf, taus)
                e1.record(main)
                e1.synchronize()
                row.append(e0.elapsed_time(e1) / reps * 1e3 - copy_us)
            px = w * h * n
            print(f"{w:>5d}x{h:<6d} {steps:>5d} {row[0]:12.1f} {row[1]:12.1f} {12.0 * px * steps / row[0] / 1e6:12.2f} {copy_us:12.1f}")
