#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 4 1 2 8 16; do echo -n "FIN_MAX=$f: G2 "; OZK_MSM_FIN_MAX=$f python tools/run_entry.py var_g2 10 2>&1 | grep -v amdgpu.ids | tail -1 | tr '\n' ' '; echo -n " G1 "; OZK_MSM_FIN_MAX=$f python tools/run_entry.py var_g1 20 2>&1 | grep -v amdgpu.ids | tail -1; done
