#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_17; mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/host_jitter.py 200 --interleaved 2>&1 | grep -v amdgpu.ids | tee $O/jitter_interleaved.txt
python tools/host_jitter.py 200 --interleaved --no-gc 2>&1 | grep -v amdgpu.ids | tee $O/jitter_interleaved_nogc.txt
