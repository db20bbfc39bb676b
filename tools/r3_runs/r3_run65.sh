#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_var_msm_gpu.py tests/test_pipeline3_gpu.py tests/test_sharded_gpu.py -x -q -m gpu > gpurun_out/t65.txt 2>&1; tail -60 gpurun_out/t65.txt | cut -c1-220
