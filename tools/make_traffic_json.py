#!/usr/bin/env python3
"""profiles/fed_pmc_traffic.json from the PMC passes collected by tools/collect_profiles.sh.
Usage: make_traffic_json.py gpurun_out <tag> <algorithmic_bytes_per_launch>"""
import collections, csv, glob, json, re, sys

out_dir, tag, algo = sys.argv[1], sys.argv[2], float(sys.argv[3])


def base(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name.split("(")[0].replace("void ", "").replace("akz::", "").strip()


def agg(pattern, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(glob.glob(pattern)[0])):
        if r["Counter_Name"] == counter:
            k = base(r["Kernel_Name"])
            d[k][0] += 1
            d[k][1] += float(r["Counter_Value"])
    return d


f = agg(f"{out_dir}/{tag}_fetch/*/*_counter_collection.csv", "FETCH_SIZE")
w = agg(f"{out_dir}/{tag}_write/*/*_counter_collection.csv", "WRITE_SIZE")
cf = agg(f"{out_dir}/{tag}_cal_fetch/*/*_counter_collection.csv", "FETCH_SIZE")
cw = agg(f"{out_dir}/{tag}_cal_write/*/*_counter_collection.csv", "WRITE_SIZE")
fed = [k for k in f if k.startswith("k_fed_own")]
n = sum(f[k][0] for k in fed)
hbm = sum(f[k][1] + w[k][1] for k in fed) * 1024.0 / n
kib = lambda d, k: round(d[k][1] / max(1, d[k][0]), 1)
cal4k = [k for k in cf if k.startswith("k_fed_own")][0]
ncopy = cf["__amd_rocclr_copyBuffer"][0]
doc = dict(
    kernel="k_fed_own", workload="bench.py default (32 x 1920x1080 frames per step)", launches=n,
    hbm_bytes_per_launch=round(hbm), algorithmic_bytes_per_launch=round(algo),
    traffic_over_algorithmic=round(hbm / algo, 3),
    formula="(FETCH_SIZE + WRITE_SIZE) KiB x 1024, separate --pmc passes; no x2 on FETCH_SIZE for this kernel's "
            "tile loads (see calibration and profiles/README.md)",
    calibration=dict(
        copy_512MiB=dict(kernel="__amd_rocclr_copyBuffer", launches=ncopy, fetch_kib_per_launch=kib(cf, "__amd_rocclr_copyBuffer"),
                         write_kib_per_launch=kib(cw, "__amd_rocclr_copyBuffer"),
                         conclusion="WRITE_SIZE exact; FETCH_SIZE = 0.5 x bytes for a wide (1 KiB per wave) stream"),
        fed_own_4k=dict(kernel=cal4k, algorithmic_read_kib=64800, worst_case_read_kib=121500, fetch_kib_per_launch=kib(cf, cal4k),
                        write_kib_per_launch=kib(cw, cal4k),
                        conclusion="doubling FETCH_SIZE would exceed the worst case (every halo byte from HBM): the counter is "
                                   "exact for these 288-320 B row-segment loads"),
        fed_step_4k=dict(kernel="k_fed_step", algorithmic_read_kib=64800, fetch_kib_per_launch=kib(cf, "k_fed_step"))),
    per_kernel={k: dict(launches=f[k][0], fetch_kib_per_launch=kib(f, k), write_kib_per_launch=kib(w, k)) for k in f})
json.dump(doc, open("profiles/fed_pmc_traffic.json", "w"), indent=1)
print({k: doc[k] for k in ("launches", "hbm_bytes_per_launch", "traffic_over_algorithmic")})
print(doc["calibration"]["copy_512MiB"], doc["calibration"]["fed_own_4k"])
