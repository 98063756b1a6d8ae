#!/bin/bash
# A/B runs of bench.py over compile-time tuning defines (rebuilds the device objects on the GPU box for each).
#   tools/sweep_tune.sh "-DAKZ_FED_WAVES=5" "-DAKZ_FED_WAVES=6"
J='import json,sys; d=json.loads(sys.stdin.readline()); r=d["roofline"]; s=d["stage_ms_per_step"]; print("%-40s %7.0f Mpix/s %6.2f ms  fed %.3f  stages: prep %.2f fed %.2f det %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], r["frac"], s["prep"], s["fed"], s["detector"]))'
for t in "$@"; do
  rm -f akaze-rust_amd/csrc/akz_kernels.o akaze-rust_amd/csrc/akz_stencil.o akaze-rust_amd/csrc/akz_stream.o akaze-rust_amd/csrc/akz_march.o akaze-rust_amd/csrc/akz_match.o
  make -C akaze-rust_amd -j8 TUNE="$t" > /dev/null 2>&1 || { echo "build failed: $t"; continue; }
  for i in 1 2; do python bench.py --steps 20 --warmup 4 --no-cpu-baseline --no-fed4k --no-single --no-match --no-self-check $BENCH_ARGS 2>/dev/null | grep '^{' | python3 -c "$J" "[$t]"; done
done
