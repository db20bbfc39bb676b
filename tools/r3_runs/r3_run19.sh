#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_19; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_var_msm_gpu.py tests/test_fixed_base_gpu.py tests/test_full_size_gpu.py tests/test_groth16_gpu.py tests/test_repeatability_gpu.py tests/test_jni_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/summary.txt
P=$GRAFT_REPO_ROOT/octopuszk_amd/libozk_prev.so
for w in fixed_g2 var_g2; do
  for i in 1 2; do
  echo -n "prev " | tee -a $O/summary.txt; OZK_LIB_PATH=$P python tools/run_entry.py $w 10 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
  echo -n "new  " | tee -a $O/summary.txt; python tools/run_entry.py $w 10 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
  done
done
for i in 1 2; do
for m in "OZK_LIB_PATH=$P" "A=new"; do
echo -n "$m: " | tee -a $O/summary.txt
env $m python tools/groth16_prove.py 20 8 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['prove_ms_best'], j['prove_gpu_ms_all'])" | tee -a $O/summary.txt
done; done
