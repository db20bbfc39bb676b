#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $1"; python tools/sched_probe.py --sched p3 --depth 4 --fit --idle-ms 0 $1 2>&1 | grep -v amdgpu.ids | grep "K= 20\|fit"; }
run ""
run "--finish"
run "--prof 2"
run "--finish --prof 2"
run "--finish --prof 2 --gc-off"
for i in 1 2; do python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 5 2>&1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo; done
