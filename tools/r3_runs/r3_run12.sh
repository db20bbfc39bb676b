#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_12; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 $2 | " | tee -a $O/summary.txt; env $1 python tools/sched_probe.py --reps 200 ${2} 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; }
P=$GRAFT_REPO_ROOT/octopuszk_amd/libozk_prev.so
for i in 1 2 3; do
run "OZK_LIB_PATH=$P" "--sched p3 --depth 4"
run "A=new" "--sched p3 --depth 4"
done
run "OZK_LIB_PATH=$P" "--sched p2 --prof 2"
run "A=new" "--sched p2 --prof 2"
run "OZK_LIB_PATH=$P" "--sched p2 --prof 2"
run "A=new" "--sched p2 --prof 2"
timeout 300 python -m pytest tests/test_var_msm_gpu.py tests/test_pipeline3_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/summary.txt
