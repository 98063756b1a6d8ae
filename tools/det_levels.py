"""Per-level timing of the detector kernel families (op entry point, 32-frame batches): tiled / streaming pair /
fused streaming, with and without the second-derivative planes.  Run through gpurun."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import torch
import akaze_amd as A

main = torch.cuda.Stream()
with torch.cuda.stream(main):
    ctx = A.Context(0, main.cuda_stream)
    n = 32
    print(f"{'level':>12s} {'S':>2s} {'keep':>5s} " + " ".join(f"{m:>10s}" for m in ("tiled", "tiled1k", "march")))
    for (w, h) in ((1920, 1080), (960, 540), (480, 270)):
        ls = torch.rand((n, h, w), device="cuda", dtype=torch.float32)
        for S in (2, 3, 4):
            for keep in (True, False):
                row = []
                for mode in (0, 4, 5):
                    ctx.set_detector_mode(mode)
                    for _ in range(2):
                        ctx.detector_response(ls, S, keep_second=keep)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    outs = {k: torch.empty_like(ls) for k in (("Lx", "Ly", "Lxx", "Lyy", "Lxy", "Ldet") if keep else ("Lx", "Ly", "Ldet"))}
                    p = lambda k: outs[k].data_ptr() if k in outs else None
                    reps = 5
                    e0.record(main)
                    for _ in range(reps):
                        A._check(A.lib().akz_op_detector_response(ctx._h, ls.data_ptr(), S, p("Lx"), p("Ly"), p("Lxx"), p("Lyy"), p("Lxy"), p("Ldet"), w, h, n))
                    e1.record(main)
                    e1.synchronize()
                    row.append(e0.elapsed_time(e1) / reps * 1e3)
                print(f"{w:>5d}x{h:<6d} {S:>2d} {str(keep):>5s} " + " ".join(f"{t:10.1f}" for t in row))
