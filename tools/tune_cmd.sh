#!/bin/bash
# Runs a command against builds with different compile-time defines (device objects rebuilt on the GPU box):
#   tools/tune_cmd.sh "python bench.py --no-single --no-match | cut -c1-120" "-DAKZ_MARCH_PF=3" "-DAKZ_MARCH_PF=6"
CMD=$1; shift
for t in "$@"; do
  rm -f akaze-rust_amd/csrc/akz_sort.o akaze-rust_amd/csrc/akz_kernels.o akaze-rust_amd/csrc/akz_stencil.o akaze-rust_amd/csrc/akz_stream.o akaze-rust_amd/csrc/akz_march.o akaze-rust_amd/csrc/akz_match.o
  make -C akaze-rust_amd -j8 TUNE="$t" > /dev/null 2>&1 || { echo "build failed: $t"; continue; }
  echo "== $t"
  bash -c "$CMD" 2>&1 | grep -v amdgpu.ids
done
