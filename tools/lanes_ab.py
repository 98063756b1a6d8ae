#!/usr/bin/env python3
"""A stream of lone frames dealt to 1 .. 4 lanes of one context (akz_ctx_set_lanes), alternated three times; variants are
akz_debug_set_schedule settings as in tools/lone_ab.py ("name:key=value,...").
    python tools/lanes_ab.py [W H] [variants]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
dev = torch.device("cuda", 0)
frame = torch.from_numpy(A.synth_frame(W, H, 0)[None]).to(dev)
cfg = A.Config()
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream); ctx.warmup()
VARIANTS = [("default", {})]
for extra in sys.argv[3:]:
    name, _, kv = extra.partition(":")
    VARIANTS.append((name, {int(k): int(v) for k, v in (x.split("=") for x in kv.split(",") if x)}))
KEYS = sorted({k for _, d in VARIANTS for k in d})
def stream(lanes, reps=200):
    ctx.set_lanes(lanes)
    for _ in range(2 * lanes + 4): ctx.extract_begin(frame, cfg).finish().close()
    torch.cuda.synchronize(); pend = []; t = time.perf_counter()
    for _ in range(reps):
        pend.append(ctx.extract_begin(frame, cfg))
        if len(pend) >= max(lanes, 2): pend.pop(0).finish().close()
    for j in pend: j.finish().close()
    return (time.perf_counter() - t) / reps
for rnd in range(3):
    for name, d in VARIANTS:
        for k in KEYS: ctx.debug_set_schedule(k, d.get(k, 0))
        print(f"{W}x{H} {name:14s} " + "  ".join(f"{l} lanes {stream(l)*1e3:.3f} ms/frame" for l in (1, 2, 3, 4)), flush=True)
ctx.set_lanes(1)
