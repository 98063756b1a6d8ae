#!/bin/bash
# A/B runs of bench.py under different argument sets (one quoted string per run).  Through gpurun from the repo root:
#   tools/sweep_args.sh "" "--threshold 1e9" "--lean"
J='import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; s=d["stage_ms_per_step"]; print("%-40s %7.0f Mpix/s %6.2f ms  fed %.3f  kp %d stages: prep %.2f fed %.2f det %.2f nms %.2f host %.2f ori %.2f mldb %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], r["frac"], d["config"]["keypoints_per_step_rank0"], s["prep"], s["fed"], s["detector"], s["nms"], s["host_kp"], s["orient"], s["mldb"]))'
for a in "$@"; do
  python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-fed4k --no-single --no-match $a 2>/dev/null | grep '^{' | python3 -c "$J" "[$a]"
done
