#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters: pmc_kernels.py counter_collection.csv [name-substring ...]"""
import csv, sys
from collections import defaultdict
import glob, os
paths = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)) if os.path.isdir(sys.argv[1]) else [sys.argv[1]]
rows = [r for p in paths for r in csv.DictReader(open(p))]
want = sys.argv[2:]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for r in rows:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("akz::", "")
    if want and not any(w in name for w in want):
        continue
    acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[name][r["Counter_Name"]] += 1
for name in sorted(acc):
    print(name)
    for c in sorted(acc[name]):
        print(f"    {c:28s} {acc[name][c] / cnt[name][c]:16.1f}  (per dispatch, {cnt[name][c]} dispatches)")
