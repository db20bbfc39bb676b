#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3_06; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_pipeline3_gpu.py -x -q 2>&1 | tail -15 | tee -a $O/summary.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench20_$i.json 2> $O/bench20_$i.err
python -c "
import json
j=json.loads(open('$O/bench20_$i.json').read().strip().splitlines()[-1]); r=j['roofline']
print('bench default 20 steps: value=%.1f ms/step=%.4f k=%s ev=%s sclk=%s single=%s' % (j['value'], j['ms_per_step'], r['kernel_ms'], r['kernel_ms_hip_events'], r['sclk_mhz'], j['config']['single_msm_latency_ms']))" | tee -a $O/summary.txt
done
python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/bench100.json 2> $O/bench100.err
python -c "
import json
j=json.loads(open('$O/bench100.json').read().strip().splitlines()[-1]); r=j['roofline']
print('bench default 100 steps: value=%.1f ms/step=%.4f k=%s ev=%s' % (j['value'], j['ms_per_step'], r['kernel_ms'], r['kernel_ms_hip_events']))" | tee -a $O/summary.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --schedule pipeline > $O/bench20_p2.json 2> $O/bench20_p2.err
python -c "
import json
j=json.loads(open('$O/bench20_p2.json').read().strip().splitlines()[-1]); r=j['roofline']
print('bench pipeline(2-stage) 20 steps: value=%.1f ms/step=%.4f k=%s ev=%s' % (j['value'], j['ms_per_step'], r['kernel_ms'], r['kernel_ms_hip_events']))" | tee -a $O/summary.txt
ls /sys/class/drm/*/device/pp_dpm_sclk 2>&1 | head -3 | tee -a $O/summary.txt; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -5 | tee -a $O/summary.txt
