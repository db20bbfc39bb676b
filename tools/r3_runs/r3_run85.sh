#!/bin/bash
cd $GRAFT_REPO_ROOT
for l in 0 58 64 43 0; do echo -n "OZK_MSM_L1=$l: "; if [ $l = 0 ]; then python tools/groth16_prove.py 20 8 2>&1 | grep -o '"prove_gpu_ms_all": [^}]*'; else OZK_MSM_L1=$l python tools/groth16_prove.py 20 8 2>&1 | grep -o '"prove_gpu_ms_all": [^}]*'; fi; done
