#!/bin/bash
cd $GRAFT_REPO_ROOT
for ks in "7,8,6" "7,6,8" "7,8,6" "7,6,8" "5,8,8"; do echo -n "qap21 KS=$ks: "; OZK_FFT_KS=$ks python tools/run_entry.py qap21 20 2>&1 | grep -v amdgpu.ids | tail -1; done
for ks in "8,8,6" "8,6,8" "8,8,6" "8,6,8"; do echo -n "fft22 KS=$ks: "; OZK_FFT_KS=$ks python tools/run_entry.py fft22 30 2>&1 | grep -v amdgpu.ids | tail -1; done
