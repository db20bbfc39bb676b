#!/bin/bash
cd $GRAFT_REPO_ROOT
for p in 0 1 0 1; do echo -n "OZK_MSM_WSUM0_PRIO=$p: "; OZK_MSM_WSUM0_PRIO=$p python tools/sched_probe.py --reps 200 --sched p3 --depth 4 2>&1 | grep -v amdgpu.ids | tail -1; done
