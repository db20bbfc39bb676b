#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_fft_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for what in fft22 qap21; do
  for i in 1 2; do python tools/run_entry.py $what 20 2>&1 | grep -v amdgpu.ids | tail -1; done
done
OZK_FFT_EVEN_FIRST=0 python tools/run_entry.py qap21 20 2>&1 | grep -v amdgpu.ids | tail -1
