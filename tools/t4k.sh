#!/bin/bash
# 8 x 3840x2160 per step under environment overrides: tools/t4k.sh "A=1" "AKZ_AUX_PRIORITY=1" ...
J='import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; s=d["stage_ms_per_step"]; print("%-46s %7.0f Mpix/s %6.2f ms  fed %.3f  stages: prep %.2f fed %.2f det %.2f nms %.2f host %.2f ori %.2f mldb %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], r["frac"], s["prep"], s["fed"], s["detector"], s["nms"], s["host_kp"], s["orient"], s["mldb"]))'
for kv in "$@"; do
  env $kv python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-fed4k --no-single --width 3840 --height 2160 --frames 8 2>/dev/null | grep '^{' | python3 -c "$J" "4K  $kv"
  env $kv python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-fed4k --no-single 2>/dev/null | grep '^{' | python3 -c "$J" "1080p  $kv"
done
