#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "" "--last-hint" "" "--last-hint"; do echo "== $f"; python tools/sched_probe.py --sched p3 --depth 4 --fit $f 2>&1 | grep -v amdgpu.ids | grep "K= 20\|K= 10\|fit"; done
