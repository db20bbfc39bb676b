#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_16; mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/host_jitter.py 100 2>&1 | grep -v amdgpu.ids | tee $O/jitter_plain.txt
python tools/host_jitter.py 100 --interleaved 2>&1 | grep -v amdgpu.ids | tee $O/jitter_interleaved.txt
OZK_FB_HOST_RANGES=8 python tools/host_jitter.py 100 --interleaved 2>&1 | grep -v amdgpu.ids | tee $O/jitter_interleaved_r8.txt
python tools/host_jitter.py 60 --interleaved --fresh-out 2>&1 | grep -v amdgpu.ids | tee $O/jitter_interleaved_fresh.txt
