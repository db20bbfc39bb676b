#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_15; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_groth16_gpu.py tests/test_qap_witness_gpu.py -x -q 2>&1 | tail -15 | tee -a $O/summary.txt
python tools/groth16_prove.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee -a $O/summary.txt
