#!/bin/bash
cd $GRAFT_REPO_ROOT
for cfg in "1024 8" "512 8" "512 6" "2048 8" "2048 11" "1024 6" "1024 10"; do set -- $cfg; echo -n "tile $1 maxk $2: "; OZK_FFT_PLAN_CACHE=1 OZK_FFT_TILE=$1 OZK_FFT_MAXK=$2 python tools/run_entry.py fft22 20 2>&1 | grep -v amdgpu.ids | tail -1; done
