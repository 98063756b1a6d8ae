#!/usr/bin/env python3
"""Times the brute-force Hamming matcher (popcount kernel k_match and matrix-core kernel k_match_mfma, each with
merge + compaction) on device-resident descriptor rows:
a 3840x2160 synthetic pair (BASELINE configs[2]) and a gathered multi-frame set."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
ctx = A.Context(0, torch.cuda.current_stream().cuda_stream)
f0, f1 = A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))
r0, r1 = ctx.extract_features(f0, keep_all_planes=False), ctx.extract_features(f1, keep_all_planes=False)
def rows(res):
    n = sum(res.counts(i)[1] for i in range(res.num_images))
    t = torch.empty((n, 64), dtype=torch.uint8, device="cuda"); res.copy_device_descriptors(t); return t
d0, d1 = rows(r0), rows(r1)
def timeit(a, b, reps=20):
    ctx.descriptor_match_device(a, b); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): out, cnt = ctx.descriptor_match_device(a, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    pairs = a.shape[0] * b.shape[0]
    print(f"n0={a.shape[0]} n1={b.shape[0]} matches={int(cnt.item())} {ms*1e3:.1f} us  {pairs/ms/1e6:.2f} Gpairs/s "
          f"({pairs*32/ms/1e9:.2f} T xor+popc lane-ops/s)")
big0 = torch.cat([d0] * 8); big1 = torch.cat([d1] * 8)
for mode, name in ((0, "popcount"), (1, "mfma i8"), (3, "mfma fp4")):
    ctx.set_match_mode(mode)
    print(name)
    timeit(d0[:2000], d1[:2000])
    timeit(d0, d1)
    timeit(big0, big1, reps=5)
# one query image against 16 train images: one launch per pair vs one multi-set launch
ctx.set_match_mode(1)
sets = [d1] * 16
cat, rows = torch.cat(sets), [int(s_.shape[0]) for s_ in sets]
def time_fn(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mode, name in ((1, "mfma i8"), (3, "mfma fp4")):
    ctx.set_match_mode(mode)
    t_pairs = time_fn(lambda: [ctx.descriptor_match_device(d0, s_) for s_ in sets])
    t_multi = time_fn(lambda: ctx.descriptor_match_sets_device(d0, cat, rows))
    pairs = d0.shape[0] * sum(rows)
    print(f"{name}: all-pairs step, 1 x {d0.shape[0]} queries against 16 x {rows[0]} rows: 16 pair launches {t_pairs*1e3:.0f} us "
          f"({pairs/t_pairs/1e9:.2f} T pairs/s), one multi-set launch {t_multi*1e3:.0f} us ({pairs/t_multi/1e9:.2f} T pairs/s)")
