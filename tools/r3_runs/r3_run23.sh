#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_23; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_qap_witness_gpu.py tests/test_fft_gpu.py tests/test_groth16_gpu.py tests/test_full_size_gpu.py tests/test_repeatability_gpu.py tests/test_pipeline3_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/summary.txt
for i in 1 2; do
echo -n "fold=0 " | tee -a $O/summary.txt; OZK_QAP_FOLD_SCALE=0 python tools/run_entry.py qap21 20 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
echo -n "fold=1 " | tee -a $O/summary.txt; python tools/run_entry.py qap21 20 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
python tools/run_entry.py fft22 20 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/groth16_prove.py 20 8 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['prove_ms_best'], j['prove_gpu_ms_all'])" | tee -a $O/summary.txt
