#!/bin/bash
cd $GRAFT_REPO_ROOT
for p in 0 1 0 1; do echo -n "OZK_PROVER_G2_PRIO=$p: "; OZK_PROVER_G2_PRIO=$p python tools/groth16_prove.py 20 8 2>&1 | grep -o '"prove_gpu_ms_all": [^}]*'; done
