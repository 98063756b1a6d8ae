#!/bin/bash
# Quick A/B of kernel durations inside the bench's step: rocprofv3 kernel stats of a short bench run, the kernels whose
# names match the patterns given (args: tag pattern...).  Run from the repo root through gpurun.
TAG=$1; shift
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp
B="$EXTRA --steps 20 --warmup 5 --no-cpu-baseline --no-single --no-match --no-fed4k --no-self-check --no-host-input --no-host-share-leg"
rm -rf /tmp/ks_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$TAG -- python3 $R/bench.py $B > $O/bench.log 2>&1
f=$(find /tmp/ks_$TAG -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
cp $(find /tmp/ks_$TAG -name "*kernel_trace.csv" | head -1) $O/kernel_trace.csv
python3 $R/tools/queue_busy.py $O/kernel_trace.csv > $O/queue_busy.txt
python3 $R/tools/step_timeline.py $(find /tmp/ks_$TAG -name "*kernel_trace.csv" | head -1) > $O/step_timeline.txt
python3 - "$f" "$@" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pats = sys.argv[2:]
for r in rows:
    n = r["Name"].replace("void akz::(anonymous namespace)::", "").replace("akz::(anonymous namespace)::", "").split("(")[0]
    if not pats or any(p in n for p in pats):
        print(f"{n[:52]:52s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us min {int(r['MinNs'])/1e3:8.1f}")
PY
grep -o '"value": [0-9.]*' $O/bench.log | head -1
