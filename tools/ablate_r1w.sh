#!/bin/bash
# static shares of the collision queue: strided or contiguous, sliced throughout
# (window 0) or only near the end (window 128)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in contig contig_always always; do
  export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so
  run csp 400 100000000 10 2
  run csp 400 12500000 10 2
  run scatter 400 20000000 1 2
  run split 800 20000000 1 2
done
