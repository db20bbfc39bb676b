#!/bin/bash
cd $GRAFT_REPO_ROOT
OZK_HOST_TRACE=1 python tools/host_path.py 20 > gpurun_out/host_poll.txt 2>&1
grep -v "stage_wait #" gpurun_out/host_poll.txt | cut -c1-250
echo "== blocking waits (A/B)"
OZK_HOST_BLOCKING_WAITS=1 python tools/host_path.py 20 2>&1 | grep "double\|var_msm_host" | cut -c1-250
