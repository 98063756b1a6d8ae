#!/bin/bash
# A/B runs of bench.py under environment overrides (tuning knobs read once per process).  Through gpurun from the repo root:
#   tools/sweep_env.sh "AKZ_FED_TH4=32" "AKZ_FED_TH4=48" "AKZ_PERSIST_BLOCKS=512" ...
J='import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; s=d["stage_ms_per_step"]; print("%-40s %7.0f Mpix/s %6.2f ms  fed %.3f  stages: prep %.2f fed %.2f det %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], r["frac"], s["prep"], s["fed"], s["detector"]))'
for kv in "$@"; do
  env $kv python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-fed4k --no-single --no-match 2>/dev/null | grep '^{' | python3 -c "$J" "$kv"
done
