#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/sharded_stress.py 300 2>&1 | grep -v amdgpu.ids | tail -25
