#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== polling"; python tools/host_jitter.py 150 --no-gc 2>&1 | grep -v amdgpu.ids | grep "min \|inside" | cut -c1-200
echo "== blocking"; OZK_HOST_BLOCKING_WAITS=1 python tools/host_jitter.py 150 --no-gc 2>&1 | grep -v amdgpu.ids | grep "min \|inside" | cut -c1-200
