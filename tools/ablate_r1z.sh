#!/bin/bash
# REFILL before COLLIDE once NEUTRAL_REFILL_MIN lanes are empty (instead of only when fewer than 48 lanes collide)
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in ${LIBS:-rfirst4 rfirst8 rfirst16}; do
  export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so
  run scatter 400 20000000 1 2
  run split 800 20000000 1 2
  run csp 400 100000000 10 2
  run csp 400 12500000 10 2
done
