#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_11; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 $2 | " | tee -a $O/summary.txt; env $1 python tools/sched_probe.py --reps 100 ${2} 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; }
for L in 40 44 48 52 56 60 64 72 80 84 85 86 88 96 112 128; do
run "OZK_MSM_L1=$L" "--sched p2 --prof 2"
done
for L in 40 44 48 56 64 84 85 86 88 96; do
run "OZK_MSM_L1=$L" "--sched p3 --depth 4"
done
