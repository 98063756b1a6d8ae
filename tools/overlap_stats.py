#!/usr/bin/env python3
"""How much do the kernels of concurrent chains overlap?  From a rocprofv3 kernel_trace.csv of tools/single_probe.py:
over the streamed phase (the middle half of all dispatches), the sum of kernel durations, the time at least one / at
least two kernels were running, and the dispatches per hardware queue.  Usage: overlap_stats.py kernel_trace.csv"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[n // 4: 3 * n // 4]
ev = []
queues = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1)); ev.append((e, -1))
    queues[r["Queue_Id"]] = queues.get(r["Queue_Id"], 0) + 1
ev.sort()
depth, last, busy1, busy2, hist = 0, ev[0][0], 0, 0, {}
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    hist[depth] = hist.get(depth, 0) + t - last
    depth += d; last = t
span = ev[-1][0] - ev[0][0]
total = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
frames = sum(1 for r in rows if "k_blur" in r["Kernel_Name"] and "unsigned char" in r["Kernel_Name"])
print(f"{len(rows)} dispatches, {frames} frames over {span/1e3:.0f} us ({span/1e3/max(1,frames):.1f} us per frame); "
      f"sum of kernel durations {total/1e3/max(1,frames):.1f} us per frame")
print(f"time with >= 1 kernel running {busy1/span:.2f}, >= 2 running {busy2/span:.2f}; depth histogram "
      + ", ".join(f"{k}: {v/span:.2f}" for k, v in sorted(hist.items())))
print("dispatches per queue:", queues)
