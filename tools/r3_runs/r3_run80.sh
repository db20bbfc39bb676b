#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== rounds rule (default)"; python tools/g2_sizes.py 2>&1 | grep -v amdgpu.ids
echo "== OZK_MSM_L1_ROUNDS=0"; OZK_MSM_L1_ROUNDS=0 python tools/g2_sizes.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "g2 or G2 or double or groth or full_size or sharded or repeat" 2>&1 | tail -3
