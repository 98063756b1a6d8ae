#!/usr/bin/env python3
"""Soak of the pipelined batch path: batches of 1-7 1080p frames (2-14 Mpx: both sides of the 11 Mpx threshold of the march
kernels, the fork and the run-ahead stages) from device or pinned host memory, with and without AKZ_INPUT_READY, up to
three batches in flight, the finish half on the caller's thread or on the context's own (akz_ctx_set_eager_finish), every
schedule variant of akz_debug_set_schedule; keypoints and descriptors of every frame against the synchronous extraction of that frame alone.
python tools/batch_soak.py [seconds]"""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
random.seed(11)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
NF = 12
host = [A.synth_frame(1920, 1080, 400 + i) for i in range(NF)]
with torch.cuda.stream(st):
    ref_ctx = A.Context(0, st.cuda_stream)
    want = []
    for f in host:
        r = ref_ctx.extract_features(torch.from_numpy(f[None]).to(dev), keep_all_planes=False)
        want.append((r.keypoints(0).tobytes(), r.descriptors(0).tobytes()))
        r.close()
    ctx = A.Context(0, st.cuda_stream)
    t_end = time.time() + budget
    batches = frames = 0
    inflight = []

    def check(item):
        global batches, frames
        ids, job, keep = item
        res = job.finish()
        for i, k in enumerate(ids):
            assert res.keypoints(i).tobytes() == want[k][0], ("keypoints", ids, i)
            assert res.descriptors(i).tobytes() == want[k][1], ("descriptors", ids, i)
        res.close()
        batches += 1
        frames += len(ids)

    flips = 0
    while time.time() < t_end:
        if batches >= 40 * (flips + 1):  # a new schedule every 40 batches, between jobs: the finish half on the context's own
            while inflight:              # thread or on the caller's, early stages on the copy stream or their own, gated or not
                check(inflight.pop(0))
            flips += 1
            ctx.set_eager_finish(random.random() < 0.6)
            ctx.debug_set_schedule(0, random.randrange(4))
            ctx.debug_set_schedule(1, random.randrange(2))
        k = random.randrange(0, 3)       # jobs left in flight while the next one is begun (the context holds three)
        while len(inflight) > k:
            check(inflight.pop(0))
        n = random.randrange(1, 8)
        ids = [random.randrange(NF) for _ in range(n)]
        arr = np.stack([host[k] for k in ids])
        lean = random.random() < 0.5
        if random.random() < 0.4:
            t = torch.from_numpy(arr).pin_memory()
            job = ctx.extract_begin_host(t, keep_all_planes=not lean)
        else:
            t = torch.from_numpy(arr).to(dev)
            ready = random.random() < 0.6
            if ready:
                torch.cuda.synchronize()  # the promise of AKZ_INPUT_READY: the upload is complete
            job = ctx.extract_begin(t, keep_all_planes=not lean, input_ready=ready)
        inflight.append((ids, job, t))
    while inflight:
        check(inflight.pop(0))
    ctx.close()
    ref_ctx.close()
print(f"batch soak: {batches} batches, {frames} frames checked, no mismatch")
