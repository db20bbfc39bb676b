#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 56 43 86 85 58 29 56; do echo -n "G1 OZK_MSM_L1=$l: single "; OZK_MSM_L1=$l python tools/run_entry.py var_g1 30 2>&1 | grep -v amdgpu.ids | tail -1 | tr '\n' ' '; echo -n " | pipelined "; OZK_MSM_L1=$l python tools/sched_probe.py --reps 200 --sched p3 --depth 4 --prof 2 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150; done
