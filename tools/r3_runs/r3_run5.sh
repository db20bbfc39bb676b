#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3_05; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 $2 | " | tee -a $O/summary.txt; env $1 python tools/sched_probe.py --reps 200 ${2} 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; }
run "A=0" "--sched p3 --depth 4"
run "A=0" "--sched p3 --depth 4 --prof 1"
run "A=0" "--sched p3 --depth 4 --prof 4"
run "A=0" "--sched p3 --depth 4 --prof 10"
run "A=0" "--sched p2"
run "A=0" "--sched p2 --prof 1"
run "OZK_MSM_TAIL_MODE=0" "--sched p2"
run "OZK_MSM_TAIL_MODE=0" "--sched p3 --depth 4"
run "A=0" "--sched p3 --depth 2 --tail-streams 1"
run "A=0" "--sched p3 --depth 3 --tail-streams 3"
run "A=0" "--sched p3 --depth 4 --own-sort-stream"
run "OZK_MSM_L1_MIN=48" "--sched p3 --depth 4"
run "OZK_MSM_L1_MIN=40" "--sched p3 --depth 4"
run "OZK_L1_LDS=81920" "--sched p3 --depth 4"
run "OZK_MSM_S=8" "--sched p3 --depth 4"
run "OZK_MSM_TAIL_SERIAL_ABOVE=16" "--sched p3 --depth 4"
run "OZK_MSM_TAIL_SERIAL_ABOVE=256" "--sched p3 --depth 4"
run "A=1" "--sched p3 --depth 4"
python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a $O/summary.txt
