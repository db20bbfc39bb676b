#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "" fb4 fb16 ""; do
  if [ -z "$v" ]; then unset OZK_LIB_PATH; else export OZK_LIB_PATH=$GRAFT_REPO_ROOT/octopuszk_amd/libozk_$v.so; fi
  for w in fixed_g1 fixed_g2; do echo -n "lib[$v] "; python tools/run_entry.py $w 20 2>&1 | grep -v amdgpu.ids | tail -1; done
done
