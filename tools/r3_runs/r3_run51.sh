#!/bin/bash
cd $GRAFT_REPO_ROOT
for sk in "" "var" "prep" "var,prep" "var,prep,dev"; do echo -n "skip[$sk]: "; HP_STOP=1 HP_SKIP=$sk python tools/host_path_bisect.py 20 2>&1 | grep "double" | cut -c100-200; done
