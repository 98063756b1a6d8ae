#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel.

Usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> [out.json]

HBM bytes per launch follow MI355X_MICROARCH.md (HBM / rocprofv3 section): the counters are in
KiB (x1024); on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced
streaming read, so it is doubled for kernels whose loads are float4 (k_fed_fused); WRITE_SIZE is
exact for 16 B/lane streaming stores.  For kernels with 4 B/lane accesses the read factor is
uncalibrated and both the raw and the doubled figure are printed.
"""
import collections
import csv
import json
import re
import sys


def base(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = name.split("(")[0].replace("void ", "").replace("akz::", "")
    return name.strip()


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = base(r["Kernel_Name"])
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def main():
    f = agg(sys.argv[1], "FETCH_SIZE")
    w = agg(sys.argv[2], "WRITE_SIZE")
    rows = {}
    print(f"{'kernel':44s} {'launches':>8s} {'FETCH KiB/launch':>18s} {'WRITE KiB/launch':>18s} {'HBM MB/launch (2xF+W)':>22s}")
    for k in sorted(f, key=lambda k: -f[k][1]):
        n, fs = f[k]
        wn, ws = w.get(k, [0, 0.0])
        fpl, wpl = fs / n, (ws / wn if wn else 0.0)
        hbm = (2.0 * fpl + wpl) * 1024.0
        rows[k] = dict(launches=n, fetch_kib_per_launch=fpl, write_kib_per_launch=wpl,
                       hbm_bytes_per_launch_corrected=hbm, hbm_bytes_per_launch_raw=(fpl + wpl) * 1024.0)
        print(f"{k[:44]:44s} {n:8d} {fpl:18.1f} {wpl:18.1f} {hbm / 1e6:22.2f}")
    if len(sys.argv) > 3:
        fed = {k: v for k, v in rows.items() if k.startswith("k_fed_fused")}
        n = sum(v["launches"] for v in fed.values())
        hbm = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in fed.values()) / max(1, n)
        json.dump(dict(kernel="k_fed_fused", launches=n, hbm_bytes_per_launch=round(hbm),
                       correction="FETCH_SIZE x2 (gfx950, 16 B/lane reads) + WRITE_SIZE, KiB -> bytes",
                       per_kernel=rows), open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
