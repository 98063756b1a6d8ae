#!/usr/bin/env python3
"""Timeline of the last complete step in a rocprofv3 kernel_trace.csv (start offset, duration, queue, kernel, grid):
which launch belongs to which pyramid level is readable from the order.  Usage: step_timeline.py kernel_trace.csv"""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_blur5" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    name = r["Kernel_Name"].replace("void ", "").replace("akz::(anonymous namespace)::", "").split("(")[0]
    s, d = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    gx = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{s/1e3:9.1f} {d/1e3:8.1f} q{r['Queue_Id']} {name[:44]:44s} grid=({gx},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) "
          f"vgpr={r['VGPR_Count']} lds={r['LDS_Block_Size']}")
