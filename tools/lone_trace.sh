#!/bin/bash
# A stream of lone frames under rocprofv3: the last dispatches by start time, with their queues (args: tag sched4 [W H])
tag=$1; s4=${2:-0}; W=${3:-1920}; H=${4:-1080}
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lt_$$
cat > /tmp/lt_$$.py <<PY
import os, sys, time
sys.path.insert(0, os.path.join("$GRAFT_REPO_ROOT", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
dev = torch.device("cuda", 0)
frame = torch.from_numpy(A.synth_frame($W, $H, 0)[None]).to(dev)
cfg = A.Config()
st = torch.cuda.Stream(dev); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream); ctx.warmup()
ctx.debug_set_schedule(4, $s4)
for kv in "$KEYS".split(","):
    if kv: ctx.debug_set_schedule(int(kv.split("=")[0]), int(kv.split("=")[1]))
for _ in range(10): ctx.extract_begin(frame, cfg).finish().close()
torch.cuda.synchronize()
t = time.perf_counter(); prev = None
for _ in range(60):
    j = ctx.extract_begin(frame, cfg)
    if prev is not None: prev.finish().close()
    prev = j
prev.finish().close()
print("streamed ms/frame", (time.perf_counter() - t) / 60 * 1e3)
torch.cuda.synchronize(); time.sleep(0.05)
ctx.extract_begin(frame, cfg).finish().close()   # one synchronous call last
PY
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/lt_$$ -- python3 /tmp/lt_$$.py > $O/run.log 2>&1
f=$(find /tmp/lt_$$ -name "*kernel_trace.csv" | head -1)
m=$(find /tmp/lt_$$ -name "*memory_copy_trace.csv" | head -1)
python3 - "$f" "$m" > $O/timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
try:  # copies that an SDMA engine carries do not show as kernels
    rows += [dict(r, Kernel_Name="COPY " + r.get("Direction", "").replace("MEMORY_COPY_", ""), Queue_Id="-", Workgroup_Size_X="1", Grid_Size_X="1")
             for r in csv.DictReader(open(sys.argv[2]))]
except Exception:
    pass
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("void ", "").replace("akz::(anonymous namespace)::", "").split("(")[0][:40]
blur = [i for i, r in enumerate(rows) if ("k_blur" in nm(r) or "k_head" in nm(r)) and "unsigned char" in nm(r)]  # a frame's first kernel
# the streamed part: frames -8 .. -3 ; the synchronous call: the last blur
a, b = blur[-7], blur[-3]
t0 = int(rows[a]["Start_Timestamp"])
print("--- streamed frames (4 frames)")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} q{r['Queue_Id']} {nm(r)}")
a = blur[-1]; t0 = int(rows[a]["Start_Timestamp"])
print("--- the synchronous call")
for r in rows[a:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:7.1f} q{r['Queue_Id']} {nm(r)}")
PY
cat $O/run.log | tail -2
