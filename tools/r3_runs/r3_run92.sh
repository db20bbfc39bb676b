#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_groth16_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
for p in 1 0 1 0; do echo -n "OZK_PROVER_LAST_LONE=$p: "; OZK_PROVER_LAST_LONE=$p python tools/groth16_prove.py 20 8 2>&1 | grep -o '"prove_gpu_ms_all": [^}]*'; done
