"""Single-frame extract_features calls (BASELINE configs[1]) issued from K host threads, one context + stream each:
a lone 1080p chain is launch-latency bound and leaves most of the chip idle, K concurrent chains fill it.
Run through gpurun:  python tools/single_multi_ctx.py [frames_per_call]"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np
import torch
import akaze_amd as A

W, H = 1920, 1080
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
frames = torch.from_numpy(np.stack([A.synth_frame(W, H, i) for i in range(8)])).to(dev)
cfg = A.Config()
torch.cuda.synchronize()


def worker(k, reps, out, barrier):
    st = torch.cuda.Stream(dev)
    ctx = A.Context(0, st.cuda_stream)
    one = frames[k % 8: k % 8 + 1] if F == 1 else frames[:F]
    for _ in range(5):
        ctx.extract_begin(one, cfg).finish().close()
    barrier.wait()
    t0 = time.perf_counter()
    prev = None
    for _ in range(reps):
        job = ctx.extract_begin(one, cfg)
        if prev is not None:
            prev.finish().close()
        prev = job
    prev.finish().close()
    out[k] = time.perf_counter() - t0
    barrier.wait()
    ctx.close()


for K in (1, 2, 3, 4, 6, 8):
    reps = 100
    out = [0.0] * K
    bar = threading.Barrier(K)
    th = [threading.Thread(target=worker, args=(k, reps, out, bar)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = max(out)
    print(f"K={K}: {F} frame(s) per call, {K * reps} calls in {wall * 1e3:.1f} ms -> {wall / (K * reps * F) * 1e3:.3f} ms per frame, "
          f"{W * H * F * K * reps / wall / 1e6:.0f} Mpix/s", flush=True)
