#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "bound=$1 warmup $2: "; OZK_PIPE_BOUND_HOST=$1 python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup $2 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print(j['value'], j['ms_per_step'], r.get('kernel_ms'))
"; }
run 0 5; run 1 5; run 0 50; run 1 50; run 0 5; run 1 5
