#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_20; mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/host_path.py 2>&1 | grep -v amdgpu.ids | tee $O/host_path.txt
