#!/bin/bash
# Collects the round's evidence on the GPU box: bench line, rocprofv3 kernel stats, PMC traffic passes (one counter per
# pass, --kernel-trace only, as MI355X_MICROARCH.md prescribes), the counter calibration on known byte counts and one SQ
# pass.  Run from the repo root through gpurun; outputs land in gpurun_out/$1_*; tools/make_pmc_traffic.py turns them
# into profiles/.
# BENCH_EXTRA="--frames 8 --width 3840 --height 2160 [--octaves 5 --sublevels 5]" collects the same evidence for another shape.
TAG=${1:-rXX}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp
B="$BENCH_EXTRA --regions 1 --no-cpu-baseline --no-single --no-match --no-fed4k --no-self-check --no-host-input --no-host-share-leg"
if [ -n "$BENCH_EXTRA" ]; then python3 $R/bench.py $BENCH_EXTRA --no-single --no-match --no-fed4k --no-host-share-leg 2>/dev/null | grep '^{' > $O/${TAG}_bench.json
else python3 $R/bench.py 2>/dev/null | grep '^{' > $O/${TAG}_bench.json; fi
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py $B > $O/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 $B > $O/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --steps 2 --warmup 1 $B > $O/${TAG}_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cal_fetch -- python3 $R/tools/pmc_calib.py > $O/${TAG}_cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cal_write -- python3 $R/tools/pmc_calib.py > $O/${TAG}_cal_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/${TAG}_sq -- python3 $R/bench.py --steps 2 --warmup 1 $B > $O/${TAG}_sq.log 2>&1
python3 $R/tools/pmc_sq.py $(find $O/${TAG}_sq -name "*counter_collection.csv" | head -1) > $O/${TAG}_sq_counters.txt 2>&1
# keep only the small summaries of the traces (the per-dispatch traces are tens of MB)
find $O/${TAG}_stats -name "*kernel_trace.csv" -delete; find $O/${TAG}_fetch $O/${TAG}_write $O/${TAG}_cal_fetch $O/${TAG}_cal_write $O/${TAG}_sq -name "*kernel_trace.csv" -delete
cut -c1-300 $O/${TAG}_bench.json
