#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/sharded_stress.py 300 --scramble --keep-contexts 2>&1 | grep -v amdgpu.ids | tail -15
timeout -k 10 500 python tools/sharded_stress.py 200 --scramble 2>&1 | grep -v amdgpu.ids | tail -15
