#!/usr/bin/env python3
"""Timeline of the last pair call and the last multi-set call in a kernel trace of tools/match_trace.py."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "akz" in r["Kernel_Name"] or "rocclr" in r["Kernel_Name"]]
def nm(r):
    return r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("akz::", "")
def show(sel):
    t0 = int(sel[0]["Start_Timestamp"])
    for r in sel:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:8.1f} {nm(r)[:56]:56s} wgs={int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])} x {r['Workgroup_Size_X']}")
firsts = [i for i, r in enumerate(rows) if nm(r).startswith("k_unpack_pair")]
a = firsts[-1]  # the last pair call: unpack, scan, merge + compaction
print("pair call:")
show(rows[a:a + 3])
print("multi-set call:")
show(rows[-5:])
