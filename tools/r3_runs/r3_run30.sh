#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 5 50 5 200 5; do echo -n "warmup $w: "; python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup $w 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done
