#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_18; mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in "A=0" "OZK_MSM_TAIL_MODE=1" "A=0" "OZK_MSM_TAIL_MODE=1"; do
echo -n "$m: " | tee -a $O/summary.txt
env $m python tools/groth16_prove.py 20 8 2>&1 | grep '^{' | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print(j['prove_ms_best'], j['prove_gpu_ms_all'])" | tee -a $O/summary.txt
done
for m in "A=0" "OZK_MSM_TAIL_MODE=1"; do
echo -n "$m var_g2: " | tee -a $O/summary.txt; env $m python tools/run_entry.py var_g2 10 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(j['value'], j['ms_per_step'], r['kernel_ms'], r['kernel_ms_alone'], r.get('alu_alone'), r.get('frac_alone'))" | tee -a $O/summary.txt
