#!/usr/bin/env python3
"""Lone frames (BASELINE configs[1] taken literally), A/B over schedule keys in ONE process: latency of a synchronous call,
ms per streamed frame (next frame begun before this one is finished), GPU time of the begin chain.
    python tools/lone_ab.py [W H] -- variants are akz_debug_set_schedule settings ("name:key=value,...";
key "select": akz_debug_set_select, key "libm": akz_debug_set_device_libm), alternated three times."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
dev = torch.device("cuda", 0)
NF = int(os.environ.get("LONE_FRAMES", "1"))  # frames per call (1: a lone frame)
KEEP = os.environ.get("LEAN", "0") != "1"      # LEAN=1: without Lxx, Lyy, Lxy, Lstep (AKZ_KEEP_ALL_PLANES not set)
frame = torch.from_numpy(np.stack([A.synth_frame(W, H, i) for i in range(NF)])).to(dev)
cfg = A.Config()
st = torch.cuda.Stream(dev)
torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream)
ctx.warmup()
VARIANTS = [("default", {})]
for extra in sys.argv[3:]:
    name, _, kv = extra.partition(":")
    VARIANTS.append((name, {(k if k in ("select", "libm") else int(k)): int(v) for k, v in (x.split("=") for x in kv.split(",") if x)}))
def setv(d):
    KEYS = sorted({k for _, dd in VARIANTS for k in dd if k not in ("select", "libm")})
    for k in KEYS:
        ctx.debug_set_schedule(k, d.get(k, 0))
    ctx.debug_set_select(d.get("select"))  # (akz_debug_set_select: 2 device, 1 host from the neighbour lists, 0 grids; absent: automatic)
    ctx.debug_set_device_libm(d.get("libm", 1) != 0)  # ("libm=0": angles, cosines and sines from the host's libm; default: on the device where proven)
def measure():
    for _ in range(10):
        ctx.extract_begin(frame, cfg, keep_all_planes=KEEP).finish().close()
    torch.cuda.synchronize()
    reps = 100
    for _ in range(5):
        ctx.extract_features(frame, cfg, keep_all_planes=KEEP).close()
    t = time.perf_counter()
    for _ in range(reps):
        ctx.extract_features(frame, cfg, keep_all_planes=KEEP).close()          # the synchronous entry point (akz_extract_device_*)
    lat = (time.perf_counter() - t) / reps
    t = time.perf_counter(); prev = None
    for _ in range(reps):
        j = ctx.extract_begin(frame, cfg, keep_all_planes=KEEP)
        if prev is not None: prev.finish().close()
        prev = j
    prev.finish().close()
    thr = (time.perf_counter() - t) / reps
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); a = time.perf_counter(); j = ctx.extract_begin(frame, cfg, keep_all_planes=KEEP); b = time.perf_counter(); ctx.synchronize(); c = time.perf_counter()
        ts.append((b - a, c - a)); j.finish().close()
    return lat, thr, float(np.median([x for x, _ in ts])), float(np.median([y for _, y in ts]))
ref = None
for rnd in range(3):
    for name, d in VARIANTS:
        setv(d)
        lat, thr, hb, gb = measure()
        r = ctx.extract_begin(frame, cfg, keep_all_planes=KEEP).finish()
        sig = (r.keypoints(0).tobytes(), r.descriptors(0).tobytes()); r.close()
        ref = ref or sig
        print(f"{NF}x{W}x{H} {name:22s} sync call {lat*1e3:.3f} ms  streamed {thr*1e3:.3f} ms/frame  begin: host {hb*1e3:.3f} ms, GPU idle after {gb*1e3:.3f} ms  same={sig == ref}", flush=True)
print(ctx.get_profile()["placement"])
