#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1048576 1 2 1048576 1; do echo "== OZK_MSM_L1_ROUNDS=$r"; OZK_MSM_L1_ROUNDS=$r python tools/g2_sizes.py 2>&1 | grep -v amdgpu.ids | grep "2^19\|2^20\|2^21" | tr '\n' ' '; echo; OZK_MSM_L1_ROUNDS=$r python tools/groth16_prove.py 20 6 2>&1 | grep -o '"prove_gpu_ms_all": [^}]*'; done
