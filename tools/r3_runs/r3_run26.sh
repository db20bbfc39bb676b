#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 | "; python tools/sched_probe.py --reps 200 --sched p3 --depth 4 $1 2>&1 | grep -v amdgpu.ids; }
run ""
run "--sort-prio -1"
run "--sort-prio 0"
run "--acc-prio -1"
run "--tail-prio -1"
run "--sort-prio -1 --tail-prio -1"
run "--sort-prio -1 --acc-prio -1"
run ""
