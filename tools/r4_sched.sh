#!/bin/bash
# Round-4 schedule A/B: bench.py under (environment prefix | argument set) pairs, one per line of stdin-free args:
#   tools/r4_sched.sh "ENV|ARGS" ...      e.g.  tools/r4_sched.sh "|--depth 2" "GPU_MAX_HW_QUEUES=4|--depth 2 --sched 0=1,2=0"
J='import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; h=d.get("host_input") or {}; c=d["config"]; print("%-64s %7.0f Mpix/s %6.3f ms  host_in %7.0f (%.3f)  det_frac %.3f  begin %.2f finish %.2f ms" % (sys.argv[1], d["value"], d["ms_per_step"], h.get("value",0), h.get("of_hbm_resident",0), r["frac"], c["host_ms_in_begin_per_batch"], c["host_ms_in_finish_per_batch"]), c.get("stream_placement"))'
for a in "$@"; do
  e="${a%%|*}"; g="${a#*|}"
  env $e python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-fed4k --no-single --no-match --no-host-share-leg --no-self-check $g 2>/dev/null | grep '^{' | python3 -c "$J" "[$e|$g]"
done
