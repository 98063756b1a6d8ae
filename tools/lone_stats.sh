#!/bin/bash
# rocprofv3 kernel statistics of N synchronous extract_features calls on one frame (args: tag [W H] [calls])
tag=$1; W=${2:-1920}; H=${3:-1080}; N=${4:-200}
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ls_$$
cat > /tmp/ls_$$.py <<PY
import os, sys, time
sys.path.insert(0, os.path.join("$GRAFT_REPO_ROOT", "akaze-rust_amd", "python"))
import numpy as np, torch
import akaze_amd as A
frame = torch.from_numpy(A.synth_frame($W, $H, 0)[None]).cuda()
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
ctx = A.Context(0, st.cuda_stream); ctx.warmup()
cfg = A.Config()
for _ in range(10): ctx.extract_features(frame, cfg).close()
t = time.perf_counter()
for _ in range($N): ctx.extract_features(frame, cfg).close()
print("ms per call under the profiler", (time.perf_counter() - t) / $N * 1e3)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ls_$$ -- python3 /tmp/ls_$$.py > $O/run.log 2>&1
cp $(find /tmp/ls_$$ -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
tail -1 $O/run.log | head -1; grep "ms per call" $O/run.log
