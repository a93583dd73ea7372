#!/bin/bash
# empty lanes that trigger a REFILL pass of the event-regrouped kernel (NEUTRAL_REFILL_MIN, default 8):
# refilling late keeps a wave's histories at the same collision count (coherent table probes) at the
# price of idle lanes
cd $GRAFT_REPO_ROOT
run() { timeout 300 python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
for lib in "" rf24 rf48 rf60; do
  if [ -z "$lib" ]; then unset NEUTRAL_HIP_LIB; else export NEUTRAL_HIP_LIB=neutral_amd/build/libneutral_hip_$lib.so; fi
  run scatter 400 20000000 1 2
  run split 800 20000000 1 2
  run csp 400 100000000 10 2
done
unset NEUTRAL_HIP_LIB
run scatter 400 20000000 1 0
