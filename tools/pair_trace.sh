#!/bin/bash
# BASELINE configs[2] under rocprofv3: ONE synchronous extract call on the 2 x 3840x2160 batch in pinned host memory, then
# match_features; every dispatch of the last call by start time with its queue, and the host's clocks (args: tag)
tag=$1
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt_$$
cat > /tmp/pt_$$.py <<PY
import os, sys, time
sys.path.insert(0, os.path.join("$GRAFT_REPO_ROOT", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
dev = torch.device("cuda", 0)
pair = np.stack([A.synth_frame(3840, 2160, 0), A.synth_frame(3840, 2160, 0, shift=(17, 9))])
h_pair = torch.from_numpy(pair).pin_memory()
cfg = A.Config()
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream); ctx.warmup()
def one():
    a = time.perf_counter()
    rp = ctx.extract_begin_host(h_pair, cfg, keep_all_planes=True).finish()
    k0, k1, q0, q1 = rp.keypoints(0), rp.keypoints(1), rp.descriptors(0), rp.descriptors(1)
    b = time.perf_counter()
    m = A.match_features(k0, q0, k1, q1, 0.86, 1000, 3.0, ctx=ctx)
    c = time.perf_counter()
    rp.close()
    return (b - a) * 1e3, (c - b) * 1e3
for _ in range(6): one()
ts = [one() for _ in range(10)]
print("extract ms %.3f  match_features ms %.3f" % tuple(np.median(np.array(ts), axis=0)))
ctx.set_profiling(2)
ctx.get_profile(reset=True)
for _ in range(5): one()
p = ctx.get_profile(reset=True)
print({k: round(v / 5, 3) for k, v in p.items() if isinstance(v, float)})
ctx.set_profiling(0)
torch.cuda.synchronize(); time.sleep(0.05)
one()
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/pt_$$ -- python3 /tmp/pt_$$.py > $O/run.log 2>&1
f=$(find /tmp/pt_$$ -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/pt_$$ -name "*memory_copy_trace.csv" | head -1)
python3 - "$f" "$m" > $O/timeline.txt <<'PY'
import csv, sys
rows = [dict(r, kind="k") for r in csv.DictReader(open(sys.argv[1]))]
try:
    rows += [dict(r, kind="c", Kernel_Name="COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")), Queue_Id="-") for r in csv.DictReader(open(sys.argv[2]))]
except Exception as e:
    print("no copy trace", e)
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("void ", "").replace("akz::(anonymous namespace)::", "").split("(")[0][:46]
blur = [i for i, r in enumerate(rows) if "k_blur5" in nm(r) or "k_blur<" in nm(r)]
a = blur[-1]
while a > 0 and rows[a - 1]["kind"] == "c" and int(rows[a]["Start_Timestamp"]) - int(rows[a - 1]["End_Timestamp"]) < 2000000: a -= 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.1f} q{r['Queue_Id']} {nm(r)}")
PY
tail -3 $O/run.log
