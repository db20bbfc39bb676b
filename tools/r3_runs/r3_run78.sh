#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_sharded_gpu.py tests/test_config3_gpu.py -x -q -m gpu 2>&1 | tail -5
