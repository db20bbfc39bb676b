#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 56 40 48 64 72 96 128; do echo -n "G2 OZK_MSM_L1=$l: "; OZK_MSM_L1=$l python tools/run_entry.py var_g2 10 2>&1 | grep -v amdgpu.ids | tail -1; done
for k in 16 8 32; do echo -n "G2 OZK_MSM_LK=$k: "; OZK_MSM_LK=$k python tools/run_entry.py var_g2 10 2>&1 | grep -v amdgpu.ids | tail -1; done
