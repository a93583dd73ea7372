#!/bin/bash
# workgroup size of the stream kernel: 1024 (4 waves/SIMD, 128 VGPRs), 768 (3, 168), 512 (2, 256)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" sb768 sb512; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run csp 400 100000000 3 2
  run stream 400 10000000 1 2
done
