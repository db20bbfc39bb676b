#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_14; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_fixed_base_gpu.py tests/test_full_size_gpu.py tests/test_groth16_gpu.py tests/test_host_mirror_gpu.py tests/test_jni_gpu.py tests/test_repeatability_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/summary.txt
P=$GRAFT_REPO_ROOT/octopuszk_amd/libozk_prev.so
for w in fixed_g1 fixed_g2 var_g2 var_g1 fft22 qap21; do
  echo -n "prev " | tee -a $O/summary.txt; OZK_LIB_PATH=$P python tools/run_entry.py $w 10 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
  echo -n "new  " | tee -a $O/summary.txt; python tools/run_entry.py $w 10 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
python tools/groth16_prove.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee -a $O/summary.txt
