#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_groth16_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
for w in fixed_g1 fixed_g1_rebuild fixed_g2 fixed_g2_rebuild; do python tools/run_entry.py $w 20 2>&1 | grep -v amdgpu.ids | tail -1; done
python tools/host_path.py 20 2>&1 | grep "fixed" | cut -c1-130
