#!/bin/bash
# The five BASELINE.json configurations at full size on one MI355X (configs 1 and 4
# in their single-GPU form), every kernel variant where it is informative.
cd $GRAFT_REPO_ROOT
run() { python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
echo "# config 1: scatter 100^2, 1e5 particles, 1 iteration"
run scatter 100 100000 1 2
echo "# config 2: stream 400^2, 1e7 particles"
for v in 0 1 2; do run stream 400 10000000 1 $v; done
echo "# config 3: scatter 400^2, 1e8 particles (2 iterations; every particle dies in the first)"
for v in 0 2; do run scatter 400 100000000 2 $v; done
echo "# config 4 on one GPU: csp 400^2, 1e8 particles, 10 iterations"
for v in 0 1 2; do run csp 400 100000000 10 $v; done
echo "# config 5: split 800^2, 1e8 particles: naive over-particle (0) vs event-regrouped (1) vs tiled (2)"
for v in 0 1 2; do run split 800 100000000 1 $v; done
