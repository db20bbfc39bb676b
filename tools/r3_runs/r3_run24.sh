#!/bin/bash
cd $GRAFT_REPO_ROOT
P=$GRAFT_REPO_ROOT/octopuszk_amd/libozk_prev.so
for i in 1 2; do
for w in fft22 qap21; do
echo -n "prev " ; OZK_LIB_PATH=$P python tools/run_entry.py $w 20 2>&1 | grep -v amdgpu.ids
echo -n "new  " ; python tools/run_entry.py $w 20 2>&1 | grep -v amdgpu.ids
done; done
