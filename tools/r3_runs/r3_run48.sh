#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do python tools/host_path.py 20 2>&1 | grep "double\|var_msm_host G1" | cut -c1-46,100-200; done
