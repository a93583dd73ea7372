#!/bin/bash
# ablation A: cost of the tally atomics (timing-only builds)
cd $GRAFT_REPO_ROOT
for lib in "" neutral_amd/build/libneutral_hip_notally.so neutral_amd/build/libneutral_hip_plaintally.so; do
  export NEUTRAL_HIP_LIB=$lib
  [ -z "$lib" ] && unset NEUTRAL_HIP_LIB
  python tools/ablate.py stream 400 10000000 1
  python tools/ablate.py csp 400 10000000 10
  python tools/ablate.py scatter 400 2000000 1
  python tools/ablate.py split 800 2000000 1
done
