#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== S_LAT=8 (default)"; python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids | cut -c1-110
echo "== S_LAT=4"; OZK_MSM_S_LAT=4 python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids | cut -c1-110
timeout -k 10 900 python -m pytest tests/test_var_msm_gpu.py tests/test_pipeline3_gpu.py tests/test_sharded_gpu.py -x -q -m gpu 2>&1 | tail -3
