#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t82.txt 2>&1; tail -5 gpurun_out/t82.txt | cut -c1-200
python tools/g2_sizes.py 2>&1 | grep -v amdgpu.ids | tr '\n' ' '; echo
python tools/run_entry.py var_g2 10 2>&1 | grep -v amdgpu.ids | tail -1
