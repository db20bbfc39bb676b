#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t84.txt 2>&1; tail -5 gpurun_out/t84.txt | cut -c1-200
python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids | tail -9 | cut -c1-60
python tools/host_path.py 20 2>&1 | grep "var_msm" | cut -c1-46,62-100
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(j['value'], j['ms_per_step'], 'single', j['config'].get('single_msm_latency_ms'), 'alone', (r.get('kernel_ms_alone') or {}).get('median'))
"
