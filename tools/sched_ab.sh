#!/bin/bash
# Alternating bench.py runs over schedule variants on ONE box (args: tag rounds "name:--sched k=v,..." ...); prints value per run
tag=$1; rounds=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
B="--no-cpu-baseline --no-single --no-match --no-fed4k --no-self-check --no-host-input --no-host-share-leg --regions 5"
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    name=${v%%:*}; args=${v#*:}
    timeout -k 5 150 python bench.py $B $args 2>/dev/null | grep '^{' > $O/${name}_$r.json
    python3 - $O/${name}_$r.json "$name" $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(f"{sys.argv[2]:24s} round {sys.argv[3]}: {d['value']:9.1f} Mpix/s  regions {d['config']['regions']['Mpix_s']}  det frac {d['roofline']['frac']}  det us {d['roofline']['avg_launch_us']}  fed ms/step {d['roofline_2']['ms_per_step']}")
PY
  done
done
