#!/bin/bash
cd $GRAFT_REPO_ROOT
for S in 4 8 16 4 8 16; do echo -n "G1 OZK_MSM_S=$S: single "; OZK_MSM_S=$S python tools/run_entry.py var_g1 30 2>&1 | grep -v amdgpu.ids | tail -1 | tr '\n' ' '; echo -n " | pipelined "; OZK_MSM_S=$S python tools/sched_probe.py --reps 200 --sched p3 --depth 4 2>&1 | grep -v amdgpu.ids | tail -1; done
