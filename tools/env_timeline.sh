#!/bin/bash
# Per-kernel GPU time of one 32-frame batch (tools/kp_probe.py under rocprofv3 --kernel-trace) under environment
# overrides:  tools/env_timeline.sh "AKZ_MARCH_FILL=4" "AKZ_MARCH_FILL=6 AKZ_MARCH_MIN_ROWS=32"   (through gpurun)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for kv in "$@"; do
  echo "== $kv"
  rm -rf /tmp/kp_tr; (cd /tmp && env $kv rocprofv3 --kernel-trace --output-format csv -d /tmp/kp_tr -- python3 $R/tools/kp_probe.py > /dev/null 2>&1)
  python3 $R/tools/batch_timeline.py /tmp/kp_tr | grep -E "${FILTER:-SUM}"
done
