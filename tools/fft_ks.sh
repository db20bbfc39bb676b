#!/bin/bash
# pass plans of the 2^22 transform on one box (OZK_FFT_KS = stages per pass)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for ks in "8,8,6" "8,6,8" "6,8,8" "8,8,6" "8,6,8" "6,8,8" "10,6,6" "8,4,10" "10,4,8"; do
  echo "== OZK_FFT_KS=$ks: $(OZK_FFT_KS=$ks python3 $R/tools/run_entry.py fft22 40 2>&1 | tail -1)   $(OZK_FFT_KS=$ks python3 $R/tools/run_entry.py fft22 40 2>&1 | tail -1)"
done
