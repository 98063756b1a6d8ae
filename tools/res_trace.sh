#!/bin/bash
# k_octave_resident alone: duration of the launch in a lone 1080p frame's chain under rocprofv3 (args: tag [env assignments])
tag=$1; shift
O=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
rm -rf /tmp/sp_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/sp_$tag -- python3 $GRAFT_REPO_ROOT/tools/single_probe.py > $O/sp.log 2>&1
f=$(find /tmp/sp_$tag -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/chain_trace.py $f > $O/chain.txt
python3 $GRAFT_REPO_ROOT/tools/overlap_stats.py $f > $O/overlap.txt
cat $O/overlap.txt
grep -h "resident\|chain end\|streamed\|begin alone" $O/chain.txt $O/sp.log
