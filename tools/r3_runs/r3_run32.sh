#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 5 50 5 50; do echo -n "warmup $w: "; python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup $w 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(j['value'], j['ms_per_step'], {k: v for k, v in r.items() if 'kernel' in k or 'sclk' in k})
"; done
