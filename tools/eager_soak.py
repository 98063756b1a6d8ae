#!/usr/bin/env python3
"""Soak of akz_ctx_set_eager_finish: thousands of lone frames of several sizes over 2-4 lanes, jobs abandoned at random,
results freed late, other calls on the context in between, lanes re-sized; every result compared with the result of the
same frame extracted synchronously on a second context.  python tools/eager_soak.py [seconds]"""
import os, sys, time, random
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
random.seed(7)
dev = torch.device("cuda", 0)
st = torch.cuda.Stream(dev)
sizes = [(640, 360), (505, 393), (800, 450), (320, 240), (1280, 720), (1920, 1080)]
frames = [torch.from_numpy(A.synth_frame(w, h, 300 + i)[None]).to(dev) for i, (w, h) in enumerate(sizes)]
torch.cuda.synchronize()
with torch.cuda.stream(st):
    ref_ctx = A.Context(0, st.cuda_stream)
    want = []
    for f in frames:
        r = ref_ctx.extract_features(f)
        want.append((r.keypoints(0).tobytes(), r.descriptors(0).tobytes()))
    ctx = A.Context(0, st.cuda_stream)
    t_end = time.time() + budget
    done = abandoned = 0
    lanes = 3
    ctx.set_lanes(lanes)
    ctx.set_eager_finish(True)
    pending, held = [], []
    while time.time() < t_end:
        k = random.randrange(len(frames))
        pending.append((k, ctx.extract_begin(frames[k])))
        if len(pending) >= lanes + random.randrange(0, 2):
            kk, job = pending.pop(0)
            if random.random() < 0.05:
                job.abandon()
                abandoned += 1
            else:
                res = job.finish()
                assert res.keypoints(0).tobytes() == want[kk][0], ("keypoints", kk, done)
                assert res.descriptors(0).tobytes() == want[kk][1], ("descriptors", kk, done)
                held.append(res)
                done += 1
        if len(held) > 6:
            for r in held[:4]:
                r.close()
            held = held[4:]
        if random.random() < 0.01:
            ctx.synchronize()
        if random.random() < 0.004:  # re-size the lanes with everything finished
            for kk, job in pending:
                res = job.finish()
                assert res.keypoints(0).tobytes() == want[kk][0]
                res.close()
                done += 1
            pending = []
            lanes = random.choice((2, 3, 4))
            ctx.set_lanes(lanes)
            ctx.set_eager_finish(random.random() < 0.8)
    for kk, job in pending:
        job.finish().close()
    ctx.close()
    ref_ctx.close()
print(f"eager soak: {done} frames finished and checked, {abandoned} abandoned, no mismatch")
