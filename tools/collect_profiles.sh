#!/bin/bash
# Collects the round's evidence on the GPU box: bench line, rocprofv3 kernel stats, PMC traffic passes and
# the counter calibration.  Run from the repo root through gpurun; outputs land in gpurun_out/$1_*.
TAG=${1:-rXX}
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp
python3 $R/bench.py 2>/dev/null | grep '^{' > $O/${TAG}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --no-single --no-match > $O/${TAG}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fed4k --no-single --no-match > $O/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fed4k --no-single --no-match > $O/${TAG}_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cal_fetch -- python3 $R/tools/pmc_calib.py > $O/${TAG}_cal_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${TAG}_cal_write -- python3 $R/tools/pmc_calib.py > $O/${TAG}_cal_write.log 2>&1
cut -c1-400 $O/${TAG}_bench.json
