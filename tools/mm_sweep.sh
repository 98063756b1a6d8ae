#!/bin/bash
# A/B runs of tools/match_bench.py over compile-time defines of akz_match.hip (rebuilt on the GPU box):
#   tools/mm_sweep.sh "-DAKZ_MM_AHEAD=0" "-DAKZ_MM_AHEAD=3"
for t in "$@"; do
  rm -f akaze-rust_amd/csrc/akz_match.o
  make -C akaze-rust_amd -j8 TUNE="$t" > /dev/null 2>&1 || { echo "build failed: $t"; continue; }
  echo "== $t"
  python tools/match_bench.py 2>&1 | grep -v amdgpu | tail -3
done
