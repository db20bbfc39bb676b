#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 41216 1024 41216 1024 32768; do echo -n "OZK_L1_LDS=$l: "; OZK_L1_LDS=$l python tools/run_entry.py var_g1 30 2>&1 | grep -v amdgpu.ids | tail -1; done
