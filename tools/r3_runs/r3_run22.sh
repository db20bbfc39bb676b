#!/bin/bash
cd $GRAFT_REPO_ROOT
OZK_HOST_TRACE=1 python tools/host_path.py 2>&1 | grep -v amdgpu.ids | sed -n 1,30p
