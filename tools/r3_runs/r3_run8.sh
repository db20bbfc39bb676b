#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_08; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_pipeline3_gpu.py -x -q -k device_clock 2>&1 | tail -30 | tee -a $O/summary.txt
