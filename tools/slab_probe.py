"""The detector op timed on the planes of a finished 32-frame extraction (the product's slab).
'same planes': the launch is repeated on one level's planes -- its 265 MB input then comes largely out of the 256 MB
Infinity Cache.  'levels in turn': every launch reads another level's Lsmooth, as in the pyramid -- the HBM figure.
SLAB_SHORT=1 prints only the second kind (used by the compile-time sweeps)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
main = torch.cuda.Stream()
w, h, n = 1920, 1080, 32
short = bool(os.environ.get("SLAB_SHORT"))
with torch.cuda.stream(main):
    ctx = A.Context(0, main.cuda_stream)
    frames = torch.from_numpy(np.stack([A.synth_frame(w, h, i) for i in range(n)])).cuda()
    r = ctx.extract_begin(frames, keep_all_planes=True, host_descriptors=False).finish()
    def dptr(level, plane):
        p = C.c_void_p()
        A._check(A.lib().akz_result_device_plane(r._h, 0, level, plane, C.byref(p)))
        return p.value
    def mk(inp, outs, S):
        return lambda: A._check(A.lib().akz_op_detector_response(ctx._h, inp, S, outs[0], outs[1], outs[2], outs[3], outs[4], outs[5], w, h, n))
    def timed(label, calls, reps=5):
        for c in calls:
            c()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(reps):
            for c in calls:
                c()
        e1.record(main)
        e1.synchronize()
        us = e0.elapsed_time(e1) / (reps * len(calls)) * 1e3
        print(f"{label:52s} {us:7.1f} us per launch  {28 * w * h * n / us / 1e6:5.2f} TB/s", flush=True)
    P = [[dptr(lvl, k) for k in (1, 2, 3, 4, 5, 6, 9)] for lvl in range(4)]  # Lsmooth, Lx, Ly, Lxx, Lyy, Lxy, Ldet
    for S in (2, 3, 4):
        if not short:
            timed(f"S{S} same planes (level 1)", [mk(P[1][0], P[1][1:], S)], reps=10)
            timed(f"S{S} same input, outputs of levels 0..3 in turn", [mk(P[0][0], P[i][1:], S) for i in range(4)])
            timed(f"S{S} inputs of levels 0..3 in turn, same outputs", [mk(P[i][0], P[0][1:], S) for i in range(4)])
        timed(f"S{S} levels 0..3 in turn", [mk(P[i][0], P[i][1:], S) for i in range(4)])
