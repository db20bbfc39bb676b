#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "" "ws" "handle" "torchwork" "ws handle torchwork" "ws handle torchwork reuse" "ws handle torchwork sleep"; do python tools/double_stall_probe.py $f 2>&1 | grep -v amdgpu.ids | tail -1; done
