#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 | "; python tools/sched_probe.py --reps 200 --sched p3 --check $1 2>&1 | grep -v amdgpu.ids | tail -1; }
run "--depth 4"
run "--depth 4 --split-accum"
run "--depth 4"
run "--depth 4 --split-accum"
run "--depth 6 --tail-streams 3 --split-accum"
run "--depth 4 --split-accum --prof 2"
