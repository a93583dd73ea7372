#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids; }
for v in 2 1; do
  run csp 400 10000000 10 $v
  run split 800 5000000 1 $v
  run scatter 400 5000000 1 $v
  run stream 400 10000000 1 $v
done
run csp 400 100000000 10 2
