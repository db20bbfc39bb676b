#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/host_wait_ab.py 2>&1 | grep -v amdgpu.ids
