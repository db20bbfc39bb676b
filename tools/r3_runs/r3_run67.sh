#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(j['value'], j['ms_per_step'], 'single', j['config'].get('single_msm_latency_ms'), 'alone', (r.get('kernel_ms_alone') or {}).get('median'))
"; done
timeout -k 10 600 python -m pytest tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -3
