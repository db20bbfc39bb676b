#!/bin/bash
cd $GRAFT_REPO_ROOT
OZK_PROVER_PIPE3=1 timeout -k 10 600 python -m pytest tests/test_groth16_gpu.py -x -q -m gpu 2>&1 | tail -2 || exit 1
run() { echo -n "$1: "; env $1 python tools/groth16_prove.py 20 8 2>&1 | grep -o '"prove_gpu_ms_all": [^}]*'; }
run "OZK_PROVER_PIPE3=0"
run "OZK_PROVER_PIPE3=1"
run "OZK_PROVER_PIPE3=1 GPU_MAX_HW_QUEUES=8"
run "OZK_PROVER_PIPE3=0 GPU_MAX_HW_QUEUES=8"
run "OZK_PROVER_PIPE3=1 GPU_MAX_HW_QUEUES=8"
