#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python tools/ablate.py "$@" 2>&1 | grep -v amdgpu.ids | tail -1; }
run csp 400 100000000 10 2
run csp 400 100000000 10 1
run split 800 100000000 1 1
run split 800 100000000 1 2
run scatter 400 20000000 1 2
run scatter 400 20000000 1 1
run csp 400 12500000 10 2
