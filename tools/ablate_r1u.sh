#!/bin/bash
# what the per-facet loads cost the stream kernel (timing experiments, stream deck:
# uniform density, so dropping the density reload does not change its results)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" nodens noedge noboth; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run stream 400 10000000 1 2
  run csp 400 100000000 2 2
done
