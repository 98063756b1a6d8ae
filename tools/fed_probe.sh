#!/bin/bash
# k_fed_own by image size and fused step count under rocprofv3: average launch duration per (size, steps) (args: tag)
tag=$1
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp_$$
cat > /tmp/fp_$$.py <<PY
import os, sys
sys.path.insert(0, os.path.join("$GRAFT_REPO_ROOT", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream); ctx.warmup()
rng = np.random.default_rng(1)
for (w, h) in ((240, 135), (480, 270), (960, 540), (1920, 1080)):
    lt = torch.from_numpy(rng.random((1, h, w), dtype=np.float32)).cuda()
    lf = torch.from_numpy(rng.random((1, h, w), dtype=np.float32)).cuda()
    for steps in (1, 2, 4, 6, 8, 10, 12, 14, 16):
        if steps > 8 and w > 480: continue
        for _ in range(12): ctx.fed_steps(lt.clone(), lf, [0.1] * steps)
        torch.cuda.synchronize()
        print("CFG", w, h, steps, flush=True)
PY
rocprofv3 --kernel-trace --output-format csv -d /tmp/fp_$$ -- python3 /tmp/fp_$$.py > $O/run.log 2>&1
f=$(find /tmp/fp_$$ -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$O/run.log" > $O/table.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_fed_own" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cfgs = [l.split()[1:] for l in open(sys.argv[2]) if l.startswith("CFG")]
per = 12
# the warm-up calls of Context.warmup come first: take the last len(cfgs) * per launches
rows = rows[-len(cfgs) * per:]
for i, c in enumerate(cfgs):
    rs = rows[i * per:(i + 1) * per][2:]
    d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rs)
    name = rs[0]["Kernel_Name"].split("k_fed_own")[1].split("(")[0]
    print(f"{c[0]:>5}x{c[1]:<5} steps {c[2]:>2}  median {d[len(d)//2]:6.1f} us  min {d[0]:6.1f}  {name}  grid {rs[0]['Grid_Size_X']}x{rs[0]['Grid_Size_Y']}")
PY
cat $O/table.txt
