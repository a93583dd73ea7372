#!/bin/bash
# round 5, first contact: what the box gives the CPU side, the new tests, a same-box baseline line
mkdir -p gpurun_out/r05
{ nproc; python -c "import os;print('affinity',len(os.sched_getaffinity(0)),'cpu_count',os.cpu_count())";
  cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/memory.max 2>/dev/null; grep -E "MemTotal|MemAvailable" /proc/meminfo;
  lscpu | grep -E "Model name|Socket|NUMA node\(s\)|Thread|Core"; } > gpurun_out/r05/box.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_tiled_pipeline.py tests/test_hip_parity.py -m gpu -x -q -k "weighted or closed_form or intensive or taken_over or slow_thief or known_answers" > gpurun_out/r05/new_tests.log 2>&1 || { tail -30 gpurun_out/r05/new_tests.log; exit 1; }
tail -3 gpurun_out/r05/new_tests.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05/bench_base_driver.json 2> gpurun_out/r05/bench_base_driver.err || { tail gpurun_out/r05/bench_base_driver.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05/bench_base_driver.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], {k['name'][:16]: round(k['ms_per_launch'], 2) for k in d['kernels']})
PY
timeout -k 10 1100 python -m pytest tests/test_full_scale.py -m gpu -x -q -rs > gpurun_out/r05/full_scale.log 2>&1; echo "full scale rc $?"; tail -5 gpurun_out/r05/full_scale.log
