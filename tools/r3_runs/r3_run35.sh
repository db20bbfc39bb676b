#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(j['value'], j['ms_per_step'], 'single', j.get('single_msm_latency_ms'), 'L1', r.get('kernel_ms', {}).get('mean'), 'alone', (r.get('kernel_ms_alone') or {}).get('mean'), 'events', (r.get('kernel_ms_hip_events') or {}).get('mean'))
"; done
python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 5 2>&1 | grep -o '"value": [0-9.]*'
