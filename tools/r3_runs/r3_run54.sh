#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_groth16_gpu.py tests/test_jni_gpu.py tests/test_host_mirror_gpu.py tests/test_sharded_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
python tools/host_path.py 20 > gpurun_out/host_path_new.txt 2>&1; cut -c1-200 gpurun_out/host_path_new.txt | grep -v amdgpu
echo "== --pageable-uploads"; python tools/host_path.py 20 --pageable-uploads 2>&1 | grep "double" | cut -c1-200
echo "== --pageable-uploads"; python tools/host_path.py 20 --pageable-uploads 2>&1 | grep "double" | cut -c1-200
