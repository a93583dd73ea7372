#!/bin/bash
# What SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE count, on kernels whose
# vector-issue load is known by construction (tools/micro/pmc_calibration.hip): plain run, kernel trace,
# one PMC pass.  Results: gpurun_out/<tag>/pmc_calibration.log
R=$GRAFT_REPO_ROOT; tag=${1:-r05}; out=$R/gpurun_out/$tag/pmc_calibration; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B=$R/tools/micro/build/pmc_calibration
$B > $out/plain.txt 2>&1
rocprofv3 --kernel-trace --stats -d $out/ktrace --output-format csv -- $B > $out/ktrace.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM -d $out/pmc --output-format csv -- $B > $out/pmc.txt 2>&1
python3 - $out <<'PY' | tee $out/../pmc_calibration.log
import csv, glob, sys, collections
out = sys.argv[1]
print(open(out + '/plain.txt').read())
print("under rocprofv3 --pmc (the same binary; two launches per case, the second is the timed one above):")
print(open(out + '/pmc.txt').read().split('\n\n')[0][-1500:])
rows = collections.defaultdict(dict)
for f in glob.glob(out + '/pmc/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows[(int(r['Dispatch_Id']), r['Kernel_Name'].split('(')[0], int(r['Grid_Size']))][r['Counter_Name']] = float(r['Counter_Value'])
for (d, name, grid), c in sorted(rows.items()):
    valu, act, busy, wave, gui = (c.get(k, 0.0) for k in ('SQ_INSTS_VALU', 'SQ_ACTIVE_INST_VALU', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE'))
    print(f"dispatch {d:2d} {name[:22]:22s} grid {grid:8d}  INSTS_VALU {valu:.4e}  ACTIVE_INST_VALU {act:.4e} ({act / valu if valu else 0:.3f} per inst)  "
          f"BUSY_CYCLES {busy:.4e}  WAVE_CYCLES {wave:.4e}  GUI_ACTIVE {gui:.4e}  "
          f"4 x ACTIVE / (1024 x BUSY / 32) = {4 * act / (1024 * busy / 32) if busy else 0:.3f}  "
          f"4 x ACTIVE / (1024 x GUI / 8) = {4 * act / (1024 * gui / 8) if gui else 0:.3f}  BUSY/32 : GUI/8 = {busy / 32 / (gui / 8) if gui else 0:.3f}")
PY
