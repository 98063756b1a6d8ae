#!/bin/bash
# Alternating bench.py runs on ONE box, printing the legs of the reference's call shape (lone frames, lanes, contexts, the 4K pair)
# instead of the headline (args: tag rounds "name:bench args" ...)
tag=$1; rounds=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
B="--no-cpu-baseline --no-fed4k --no-self-check --no-host-input --no-host-share-leg --regions 1 --steps 10 --warmup 4"
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    name=${v%%:*}; args=${v#*:}
    timeout -k 5 200 python bench.py $B $args 2>/dev/null | grep '^{' > $O/${name}_$r.json
    python3 - $O/${name}_$r.json "$name" $r <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
s, p = d["single_frame"], d["pair_4k"]
print(f"{sys.argv[2]:14s} {sys.argv[3]}: 1080p sync {s['latency_ms']:.3f} stream {s['stream_ms_per_frame']:.3f} lanes2/3/4 "
      f"{s['lanes']['2']['stream_ms_per_frame']:.3f}/{s['lanes']['3']['stream_ms_per_frame']:.3f}/{s['lanes']['4']['stream_ms_per_frame']:.3f} "
      f"ctx4 {s['multi_context']['ms_per_frame']:.3f} | 4K sync {s['lone_4k']['sync_call_ms']:.3f} stream {s['lone_4k']['stream_ms_per_frame']:.3f} "
      f"ahead2 {s['lone_4k']['two_begun_ahead']['stream_ms_per_frame']:.3f} | pair {p['ms_per_pair']:.3f} (extract {p['extract_ms']:.3f}) streamed {p['streamed_ms_per_pair']:.3f}", flush=True)
PY
  done
done
