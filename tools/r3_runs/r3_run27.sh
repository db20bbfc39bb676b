#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/sched_probe.py --sched p3 --depth 4 --fit 2>&1 | grep -v amdgpu.ids
