#!/bin/bash
cd $GRAFT_REPO_ROOT
for idle in 0 50 500; do echo "== idle $idle ms then 5 warm-up MSMs"; python tools/sched_probe.py --sched p3 --depth 4 --fit --idle-ms $idle 2>&1 | grep -v amdgpu.ids | grep "K= 20\|fit"; done
for i in 1 2 3; do python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 5 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done
