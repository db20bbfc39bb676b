#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/idle_wake_probe.py 2>&1 | grep -v amdgpu.ids
