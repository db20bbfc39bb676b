#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_13; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 $2 | " | tee -a $O/summary.txt; env $1 python tools/sched_probe.py --reps 100 ${2} 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; }
for L in 56 92 100 104 108 116 120 136 152 168 184; do
run "OZK_MSM_L1=$L" "--sched p3 --depth 4"
done
for L in 92 100 104 108 116 120 136 152 168 184; do
run "OZK_MSM_L1=$L" "--sched p2 --prof 2"
done
