#!/bin/bash
cd $GRAFT_REPO_ROOT
for h in 1 0 1 0; do echo "== OZK_MSM_HEAD_LONE=$h"; OZK_MSM_HEAD_LONE=$h python bench.py --no-cpu-baseline --timed-only --steps 100 --warmup 5 --schedule pipeline 2>&1 | grep -o '"value": [0-9.]*'; OZK_MSM_HEAD_LONE=$h python tools/groth16_prove.py 20 8 2>&1 | grep -o '"prove_gpu_ms_all": [^}]*'; done
