#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_fixed_base_gpu.py tests/test_host_mirror_gpu.py tests/test_jni_gpu.py tests/test_groth16_gpu.py -x -q -m gpu 2>&1 | tail -15
