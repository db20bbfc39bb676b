#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo -n "default:           "; HP_STOP=1 HP_SKIP=var,prep python tools/host_path_bisect.py 20 2>&1 | grep "double" | cut -c100-200
echo -n "HSA_ENABLE_SDMA=0: "; HSA_ENABLE_SDMA=0 HP_STOP=1 HP_SKIP=var,prep python tools/host_path_bisect.py 20 2>&1 | grep "double" | cut -c100-200
done
