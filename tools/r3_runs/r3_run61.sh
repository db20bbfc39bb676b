#!/bin/bash
cd $GRAFT_REPO_ROOT
for S in 4 2 8 1; do for m in 0 1; do echo -n "G2 OZK_MSM_S=$S TAIL_MODE=$m: "; OZK_MSM_S=$S OZK_MSM_TAIL_MODE=$m python tools/run_entry.py var_g2 10 2>&1 | grep -v amdgpu.ids | tail -1; done; done
for S in 4 2 8; do echo -n "G1 OZK_MSM_S=$S: "; OZK_MSM_S=$S python tools/run_entry.py var_g1 20 2>&1 | grep -v amdgpu.ids | tail -1; done
