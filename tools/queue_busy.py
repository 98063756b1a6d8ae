#!/usr/bin/env python3
"""Per hardware queue: busy fraction, dispatches and the largest idle gaps over the steady-state part of a rocprofv3
kernel_trace.csv of bench.py (the timed steps: the dispatches after the last idle period of 20 ms or more, less the
first and last tenth).  Usage: queue_busy.py kernel_trace.csv"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
start = 0
end_so_far = 0
for i, r in enumerate(rows):  # the last long idle period of the whole device: warm-up / placement probe | timed steps
    if i and int(r["Start_Timestamp"]) - end_so_far > 20_000_000:
        start = i
    end_so_far = max(end_so_far, int(r["End_Timestamp"]))
rows = rows[start:]
n = len(rows)
rows = rows[n // 10: n - n // 10]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
def nm(r): return r["Kernel_Name"].replace("void ", "").replace("akz::(anonymous namespace)::", "").split("(")[0][:40]
byq = {}
for r in rows:
    byq.setdefault(r["Queue_Id"], []).append(r)
print(f"window {(t1 - t0) / 1e3:.0f} us, {len(rows)} dispatches")
for q, rs in sorted(byq.items()):
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    gaps = []
    for a, b in zip(rs, rs[1:]):
        g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
        if g > 0:
            gaps.append((g, nm(a), nm(b)))
    gaps.sort(reverse=True)
    print(f"queue {q}: {len(rs)} dispatches, busy {busy / (t1 - t0):.2f}; idle gaps > 20 us: "
          f"{sum(1 for g in gaps if g[0] > 20000)} totalling {sum(g[0] for g in gaps if g[0] > 20000) / 1e3:.0f} us")
    for g, a, b in gaps[:6]:
        print(f"    {g / 1e3:7.1f} us between {a} and {b}")
