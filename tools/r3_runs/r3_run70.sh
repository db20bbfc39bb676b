#!/bin/bash
cd $GRAFT_REPO_ROOT
for ks in "8,8,6" "8,6,8" "6,8,8" "8,7,7" "7,7,8" "7,8,7" "6,6,10" "10,6,6" "8,4,10" "4,8,10"; do echo -n "KS=$ks: "; OZK_FFT_KS=$ks python tools/run_entry.py fft22 20 2>&1 | grep -v amdgpu.ids | tail -1; done
