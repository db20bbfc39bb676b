#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t95.txt 2>&1; tail -4 gpurun_out/t95.txt | cut -c1-200
python tools/groth16_prove.py 20 10 2>&1 | grep -o '"prove_ms_best.*'
