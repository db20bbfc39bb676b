#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3_07; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 $2 | " | tee -a $O/summary.txt; env $1 python tools/sched_probe.py --reps 200 ${2} 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; }
timeout 300 python -m pytest tests/test_var_msm_gpu.py tests/test_pipeline3_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/summary.txt
for i in 1 2 3; do
run "OZK_L1_LAZY=0" "--sched p3 --depth 4"
run "OZK_L1_LAZY=1" "--sched p3 --depth 4"
done
run "OZK_L1_LAZY=0" "--sched p2 --prof 2"
run "OZK_L1_LAZY=1" "--sched p2 --prof 2"
run "OZK_L1_LAZY=0" "--sched p2 --prof 2"
run "OZK_L1_LAZY=1" "--sched p2 --prof 2"
