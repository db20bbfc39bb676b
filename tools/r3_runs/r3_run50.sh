#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in "realg2" "realg2 ws handle" "realg2" "realg2 reuse" "realg2 sleep"; do python tools/double_stall_probe.py $f 2>&1 | grep -v amdgpu.ids | tail -1; done
python tools/host_path.py 20 2>&1 | grep "double" | cut -c1-46,100-200
