#!/bin/bash
# Per-launch timeline of one 32-frame batch (tools/kp_probe.py under rocprofv3 --kernel-trace) for builds with the given
# compile-time defines:  tools/tune_timeline.sh "-DAKZ_EXP_NONE" "-DAKZ_EXP_NO_NMS"   (through gpurun, from the repo root)
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
for t in "$@"; do
  rm -f $R/akaze-rust_amd/csrc/akz_kernels.o $R/akaze-rust_amd/csrc/akz_stencil.o $R/akaze-rust_amd/csrc/akz_stream.o $R/akaze-rust_amd/csrc/akz_march.o
  make -C $R/akaze-rust_amd -j8 TUNE="$t" > /dev/null 2>&1 || { echo "build failed: $t"; continue; }
  echo "== $t"
  rm -rf /tmp/kp_tr; (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kp_tr -- python3 $R/tools/kp_probe.py > /dev/null 2>&1)
  python3 $R/tools/batch_timeline.py /tmp/kp_tr | grep -E "${FILTER:-march}"
done
