#!/bin/bash
cd $GRAFT_REPO_ROOT
OZK_HOST_TRACE=1 python tools/host_path.py 20 > gpurun_out/host_trace.txt 2>&1
grep -n "stage_wait\|double\|slowest" gpurun_out/host_trace.txt | cut -c1-250
