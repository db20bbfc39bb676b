#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3_04; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 | " | tee -a $O/summary.txt; env $1 python tools/sched_probe.py --sched p3 --depth 4 --tail-streams 2 --reps 200 ${2} 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; }
run "A=0"
run "OZK_MSM_S=8"
run "OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=64"
run "OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=64 OZK_L1_LDS=81920"
run "OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=64 OZK_MSM_L1_MIN=48"
run "OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=64 OZK_L1_LDS=81920 OZK_MSM_L1_MIN=48"
run "OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=8"
run "OZK_MSM_S=4 OZK_MSM_WSUM_FUSED=0 OZK_MSM_TAIL_SERIAL_ABOVE=64"
run "OZK_MSM_S=8 OZK_MSM_WSUM_FUSED=0 OZK_MSM_TAIL_SERIAL_ABOVE=64"
run "OZK_MSM_S=16 OZK_MSM_WSUM_FUSED=0 OZK_MSM_TAIL_SERIAL_ABOVE=64"
run "OZK_MSM_S=16 OZK_MSM_WSUM_FUSED=0 OZK_MSM_TAIL_SERIAL_ABOVE=64 OZK_L1_LDS=81920"
run "OZK_L1_LDS=81920"
run "OZK_L1_LDS=54784"
run "OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=64" "--depth 2 --tail-streams 1"
run "OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=64" "--depth 3 --tail-streams 3"
run "OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=64" "--prepared"
run "A=1"
