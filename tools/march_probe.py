"""Timing of the detector op entry point for one kernel family (default: the column march, det_mode 5) on 32-frame
levels; environment knobs (AKZ_MARCH_FILL, AKZ_MARCH_MIN_ROWS, AKAZE_HIP_LIB) are read by the library.  Through gpurun."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import torch
import akaze_amd as A

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 5
sizes = ((1920, 1080), (960, 540)) if len(sys.argv) < 3 else tuple(tuple(int(v) for v in a.split("x")) for a in sys.argv[2:])
main = torch.cuda.Stream()
with torch.cuda.stream(main):
    ctx = A.Context(0, main.cuda_stream)
    ctx.set_detector_mode(mode)
    n = 32
    out = []
    for (w, h) in sizes:
        ls = torch.rand((n, h, w), device="cuda", dtype=torch.float32)
        for S in (2, 3, 4):
            for keep in (True, False):
                outs = {k: torch.empty_like(ls) for k in (("Lx", "Ly", "Lxx", "Lyy", "Lxy", "Ldet") if keep else ("Lx", "Ly", "Ldet"))}
                p = lambda k: outs[k].data_ptr() if k in outs else None
                call = lambda: A._check(A.lib().akz_op_detector_response(ctx._h, ls.data_ptr(), S, p("Lx"), p("Ly"), p("Lxx"), p("Lyy"), p("Lxy"), p("Ldet"), w, h, n))
                for _ in range(2):
                    call()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 5
                e0.record(main)
                for _ in range(reps):
                    call()
                e1.record(main)
                e1.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                gb = (4 + 12 + (12 if keep else 0)) * w * h * n / 1e9
                out.append(f"{w}x{h} S{S} {'keep' if keep else 'lean'} {us:7.1f} us {gb / us * 1e6 / 1e3:5.2f} TB/s")
    tag = " ".join(f"{k}={os.environ[k]}" for k in ("AKZ_MARCH_FILL", "AKZ_MARCH_MIN_ROWS", "AKAZE_HIP_LIB") if k in os.environ)
    print(f"mode {mode} [{tag}]\n  " + "\n  ".join(out), flush=True)
