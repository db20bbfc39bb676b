#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python tools/groth16_prove.py 20 10 2>&1 | grep -o '"prove_gpu_ms_all": [^}]*'; }
run "OZK_PROVER_PIPE3=1"
run "OZK_PROVER_PIPE3=0"
run "OZK_PROVER_PIPE3=1 OZK_PROVER_TAIL_STREAMS=1"
run "OZK_PROVER_PIPE3=1"
run "OZK_PROVER_PIPE3=0"
run "OZK_PROVER_PIPE3=1 OZK_PROVER_TAIL_STREAMS=1"
