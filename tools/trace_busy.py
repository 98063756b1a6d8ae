#!/usr/bin/env python3
"""GPU busy time from a rocprofv3 kernel_trace.csv: union of kernel intervals, per-queue counts, idle gaps.
Usage: trace_busy.py kernel_trace.csv [skip_fraction]"""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_begin = iv[0][0] + (iv[-1][1] - iv[0][0]) * skip   # look at the tail (steady state)
iv = [x for x in iv if x[0] >= t_begin]
span = iv[-1][1] - iv[0][0]
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
gaps = []
for s, e, _, _ in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ksum = sum(e - s for s, e, _, _ in iv)
print(f"kernels {len(iv)}  span {span/1e6:.2f} ms  busy(union) {busy/1e6:.2f} ms ({100*busy/span:.0f}%)  "
      f"sum of kernel durations {ksum/1e6:.2f} ms  queues {dict(Counter(q for *_, q in iv))}")
gaps.sort(reverse=True)
print("largest idle gaps (ms):", [round(g / 1e6, 3) for g in gaps[:12]], " total idle", round(sum(gaps) / 1e6, 2))
dur = Counter()
for s, e, k, _ in iv:
    dur[k.split("(")[0].replace("void ", "").replace("akz::(anonymous namespace)::", "")] += e - s
for k, v in dur.most_common(12):
    print(f"  {k[:50]:50s} {v/1e6:8.2f} ms")
