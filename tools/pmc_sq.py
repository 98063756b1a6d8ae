#!/usr/bin/env python3
"""Per-kernel SQ counter ratios from a rocprofv3 --pmc counter_collection.csv (wave cycles split into active / parked /
issue-stalled, VALU and LDS activity).  Usage: pmc_sq.py counter_collection.csv"""
import collections, csv, re, sys

d = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "").replace("akz::", "").strip()
    d[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
names = sorted({c for v in d.values() for c in v})
print("counters:", names)
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = v.get("SQ_WAVE_CYCLES", 0)
    if wc <= 0:
        continue
    f = lambda c: 100.0 * v.get(c, 0) / wc
    print(f"{k[:44]:44s} launches {n[(k, 'SQ_WAVE_CYCLES')]:4d} wave_cyc {wc:.3g}  active_any {f('SQ_ACTIVE_INST_ANY'):5.1f}%  "
          f"wait_any {f('SQ_WAIT_ANY'):5.1f}%  wait_inst_any {f('SQ_WAIT_INST_ANY'):5.1f}%  valu {f('SQ_ACTIVE_INST_VALU'):5.1f}%  "
          f"lds {f('SQ_ACTIVE_INST_LDS'):5.1f}%  wait_inst_lds {f('SQ_WAIT_INST_LDS'):5.1f}%  "
          f"busy_cyc {v.get('SQ_BUSY_CYCLES', 0):.3g} insts_valu/wave_cyc {v.get('SQ_INSTS_VALU', 0) / wc:.3f}")
