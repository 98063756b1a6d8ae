#!/usr/bin/env python3
"""A stream of 8 x 3840x2160 batches, 5 octaves x 5 sublevels (BASELINE configs[4]'s extraction half), resident input,
begun two ahead: Mpix/s, for timelines (rocprofv3 --kernel-trace -- python3 tools/batch4k.py).
python tools/batch4k.py [frames per batch] [batches] [keep_all_planes 0|1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
keep = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
cfg = A.Config(num_sublevels=5, max_octave_evolution=5)
fr = torch.from_numpy(np.stack([A.synth_frame(3840, 2160, i) for i in range(n)])).cuda()
torch.cuda.synchronize()
def stream(reps, depth=2):
    pend = []
    t0 = time.perf_counter()
    for _ in range(reps):
        pend.append(ctx.extract_begin(fr, cfg, keep_all_planes=keep, host_descriptors=False, input_ready=True))
        if len(pend) > depth:
            pend.pop(0).finish().close()
    while pend:
        pend.pop(0).finish().close()
    return (time.perf_counter() - t0) / reps * 1e3
stream(4)
ms = min(stream(reps), stream(reps))
print(f"{n} x 3840x2160, 5 x 5, all planes {keep}: {ms:.3f} ms per batch ({n * 3840 * 2160 / ms / 1e3:.0f} Mpix/s)")
