"""Sensitivity of the detector march to where its 7 planes lie relative to each other: the planes are carved from one
device buffer at stride plane_bytes + PAD (PAD = 0 is the product's slab layout).  Through gpurun:
    python tools/march_layout_probe.py 0 4096 65536 1048576 torch"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import torch
import akaze_amd as A
main = torch.cuda.Stream()
w, h, n = 1920, 1080, 32
pb = w * h * n * 4
with torch.cuda.stream(main):
    ctx = A.Context(0, main.cuda_stream)
    ctx.set_detector_mode(5)
    for arg in sys.argv[1:]:
        if arg == "torch":
            planes = [torch.rand((n, h, w), device="cuda") for _ in range(7)]
            ptr = [p.data_ptr() for p in planes]
        else:
            pad = int(arg)
            big = torch.empty(7 * (pb + pad) + (1 << 21), dtype=torch.uint8, device="cuda")
            base = (big.data_ptr() + (1 << 21) - 1) >> 21 << 21
            ptr = [base + i * (pb + pad) for i in range(7)]
            src = torch.rand((n, h, w), device="cuda")
            A.copy_d2d(ptr[0], src.data_ptr(), pb)
        for S in (2, 3, 4):
            call = lambda: A._check(A.lib().akz_op_detector_response(ctx._h, ptr[0], S, ptr[1], ptr[2], ptr[3], ptr[4], ptr[5], ptr[6], w, h, n))
            for _ in range(2):
                call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            for _ in range(5):
                call()
            e1.record(main)
            e1.synchronize()
            us = e0.elapsed_time(e1) / 5 * 1e3
            print(f"layout {arg:>8s} S{S} {us:7.1f} us {28 * w * h * n / us / 1e6:5.2f} TB/s  base%2MiB={ptr[1] % (1 << 21)}", flush=True)
        planes = big = None
