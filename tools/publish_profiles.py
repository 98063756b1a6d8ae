#!/usr/bin/env python3
"""Copies the summaries of one tools/collect_profiles.sh run from gpurun_out/ into profiles/ (tracked): the bench line,
the rocprofv3 kernel statistics, per-kernel sums of every PMC pass and the SQ counter table; then regenerates
profiles/pmc_traffic.json and <tag>_pmc_per_kernel.txt (tools/make_pmc_traffic.py).  Usage: publish_profiles.py <tag>
(clean gpurun_out/<tag>_* before collecting: passes of different builds must not be mixed)."""
import collections, csv, glob, os, re, shutil, subprocess, sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    if not files:
        sys.exit(f"nothing matches {pattern}")
    return max(files, key=os.path.getmtime)


def base(name):
    return re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "").replace("akz::", "").strip()


shutil.copy(os.path.join(G, f"{tag}_bench.json"), os.path.join(P, f"{tag}_bench.json"))
shutil.copy(newest(f"{G}/{tag}_stats/**/*kernel_stats.csv"), os.path.join(P, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(G, f"{tag}_sq_counters.txt"), os.path.join(P, f"{tag}_sq_counters.txt"))
for d, out in (("fetch", "pmc_fetch_size"), ("write", "pmc_write_size"), ("cal_fetch", "pmc_cal_fetch_size"), ("cal_write", "pmc_cal_write_size")):
    files = glob.glob(f"{G}/{tag}_{d}/**/*counter_collection.csv", recursive=True)
    if len(files) != 1:
        sys.exit(f"{tag}_{d}: {len(files)} counter files (expected the one of this collection)")
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        k = (base(r["Kernel_Name"]), r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    with open(os.path.join(P, f"{tag}_{out}.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "counter", "launches", "sum", "per_launch"])
        for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            wr.writerow([k, c, n, round(s, 1), round(s / n, 1)])
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_pmc_traffic.py"), G, tag])
