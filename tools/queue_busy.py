#!/usr/bin/env python3
"""Per hardware queue: busy fraction, dispatches and the largest idle gaps over the steady-state middle of a rocprofv3
kernel_trace.csv of bench.py.  Usage: queue_busy.py kernel_trace.csv"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[n // 3: 2 * n // 3]
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
