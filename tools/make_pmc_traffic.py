#!/usr/bin/env python3
"""profiles/pmc_traffic.json (read by bench.py into roofline.traffic / traffic_frac) and profiles/<tag>_pmc_per_kernel.txt
from the PMC passes collected by tools/collect_profiles.sh.   Usage: make_pmc_traffic.py gpurun_out <tag>

HBM bytes per launch = FETCH_SIZE x fetch_factor + WRITE_SIZE, counters in KiB, one counter per pass.
MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read and other
access widths must be calibrated on a known byte count in the kernel's own access pattern.  The calibration pass
(tools/pmc_calib.py) runs the detector march on a level whose byte counts are known: the factor that makes its FETCH_SIZE
equal the known read bytes is applied to the march kernels (same access shape: 8 B per lane, 512-column strips); the
tiled kernels keep factor 1 (calibrated in round 1 on the FED kernel: doubling would exceed the worst case)."""
import collections, csv, glob, json, os, re, sys

out_dir, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def base(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return name.split("(")[0].replace("void ", "").replace("akz::", "").strip()


def agg(pattern, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                k = base(r["Kernel_Name"])
                d[k][0] += 1
                d[k][1] += float(r["Counter_Value"])
    return d


f = agg(f"{out_dir}/{tag}_fetch/**/*_counter_collection.csv", "FETCH_SIZE")
w = agg(f"{out_dir}/{tag}_write/**/*_counter_collection.csv", "WRITE_SIZE")
cf = agg(f"{out_dir}/{tag}_cal_fetch/**/*_counter_collection.csv", "FETCH_SIZE")
cw = agg(f"{out_dir}/{tag}_cal_write/**/*_counter_collection.csv", "WRITE_SIZE")
kib = lambda d, k: d[k][1] / max(1, d[k][0])

# ---- calibration ----
W, H, N = 1920, 1080, 32
plane_kib = W * H * N * 4 / 1024.0
strips, use, halo = 4, 480, 16
bands = 6
read_kib = plane_kib * (W + strips * 2 * halo) / W * (H - 6 + bands * (4 * 3 + 3)) / H  # strip halo columns x band warm-up rows (sigma 3)
cal_k = [k for k in cf if k.startswith("k_detector_march")]
cal = {}
march_factor = 1.0
if cal_k:
    k = cal_k[0]
    march_factor = read_kib / kib(cf, k)
    cal["detector_march_32x1080p"] = dict(kernel=k, known_read_kib=round(read_kib), known_write_kib=round(6 * plane_kib),
                                          fetch_kib_per_launch=round(kib(cf, k), 1), write_kib_per_launch=round(kib(cw, k), 1),
                                          fetch_factor=round(march_factor, 3),
                                          write_ratio=round(kib(cw, k) / (6 * plane_kib), 3))
copyk = "__amd_rocclr_copyBuffer"
if copyk in cf:
    cal["copy_512MiB"] = dict(kernel=copyk, fetch_kib_per_launch=round(kib(cf, copyk), 1), write_kib_per_launch=round(kib(cw, copyk), 1),
                              known_kib=512 * 1024)
fedk = [k for k in cf if k.startswith("k_fed_own")]
if fedk:
    cal["fed_own_4k"] = dict(kernel=fedk[0], algorithmic_read_kib=64800, fetch_kib_per_launch=round(kib(cf, fedk[0]), 1),
                             write_kib_per_launch=round(kib(cw, fedk[0]), 1))


def factor(k):
    return march_factor if ("_march" in k) else 1.0


def group(prefixes):
    ks = [k for k in f if any(k.startswith(p) for p in prefixes)]
    n = sum(f[k][0] for k in ks)
    hbm = sum(f[k][1] * factor(k) + w[k][1] for k in ks) * 1024.0 / max(1, n)
    return dict(launches=n, hbm_bytes_per_launch=round(hbm), kernels=ks)


lines = [f"{'kernel':44s} {'launches':>8s} {'FETCH KiB/launch':>17s} {'factor':>7s} {'WRITE KiB/launch':>17s} {'HBM MB/launch':>14s}"]
for k in sorted(f, key=lambda k: -(f[k][1] * factor(k) + w[k][1])):
    hb = (kib(f, k) * factor(k) + kib(w, k)) * 1024.0
    lines.append(f"{k[:44]:44s} {f[k][0]:8d} {kib(f, k):17.1f} {factor(k):7.3f} {kib(w, k):17.1f} {hb / 1e6:14.2f}")
open(os.path.join(ROOT, "profiles", f"{tag}_pmc_per_kernel.txt"), "w").write("\n".join(lines) + "\n")
bench = json.loads(open(f"{out_dir}/{tag}_bench.json").read().strip().splitlines()[-1])
cfgb = bench["config"]
doc = dict(
    workload=dict(width=cfgb["width"], height=cfgb["height"], frames=cfgb["frames_per_gpu"],
                  octaves=int(re.search(r"\((\d+) oct x (\d+) sub\)", bench["metric"]).group(1)),
                  sublevels=int(re.search(r"\((\d+) oct x (\d+) sub\)", bench["metric"]).group(2)),
                  lean=cfgb["planes"] == "lean"),
    source=f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 2 --warmup 1` ({tag}); "
           "HBM bytes = FETCH_SIZE x fetch_factor + WRITE_SIZE (KiB x 1024)",
    calibration=cal,
    # every kernel by its own name (bench.py attaches these to the rows of roofline.kernels / roofline_2.kernels)
    per_kernel={k: dict(launches=f[k][0], hbm_bytes_per_launch=round((kib(f, k) * factor(k) + kib(w, k)) * 1024.0))
                for k in f if k.startswith("k_")},
    kernels={
        "detector (k_detector_march + k_detector_tiled)": group(["k_detector_march", "k_detector_tiled"]),  # every detector launch, as bench.py counts them
        "k_level_march + k_fed_own": group(["k_level_march", "k_fed_own", "k_octave_resident"]),  # every diffusion launch
    })
# pmc_traffic.json is what bench.py reads for its DEFAULT workload; any other shape gets a file of its own
default_shape = (doc["workload"]["width"], doc["workload"]["height"], doc["workload"]["frames"], doc["workload"]["octaves"]) == (1920, 1080, 32, 4)
json.dump(doc, open(os.path.join(ROOT, "profiles", "pmc_traffic.json" if default_shape else f"{tag}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "kernels"} for k, v in doc["kernels"].items()}), json.dumps(cal))
print("\n".join(lines[:14]))
