#!/usr/bin/env python3
"""Launch-by-launch view of the LAST extract_begin chain in a rocprofv3 kernel_trace.csv of tools/single_probe.py:
start offset, dispatch-to-dispatch duration, gap, queue, kernel, workgroups.  Usage: chain_trace.py kernel_trace.csv [summary]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("void ", "").replace("akz::(anonymous namespace)::", "").split("(")[0]
marks = [i for i, r in enumerate(rows) if "k_blur" in nm(r) and ("unsigned char" in nm(r) or "k_blur5" in nm(r))]
a = marks[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev_end = t0
groups = {}
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    if len(sys.argv) < 3:
        print(f"{(s-t0)/1e3:8.1f} dur {(e-s)/1e3:6.1f} gap {(s-prev_end)/1e3:6.1f} q{r['Queue_Id']} {nm(r)[:44]:44s} wg={wg}")
    g = groups.setdefault(nm(r).split("<")[0], [0, 0.0]); g[0] += 1; g[1] += (e - s) / 1e3
    prev_end = max(prev_end, e)
for k, (c, t) in groups.items():
    print(f"{k:32s} {c:3d} launches {t:7.1f} us")
print(f"chain end at {(prev_end-t0)/1e3:.1f} us")
