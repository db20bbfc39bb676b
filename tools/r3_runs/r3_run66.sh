#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/t66.txt 2>&1; tail -25 gpurun_out/t66.txt | cut -c1-250
