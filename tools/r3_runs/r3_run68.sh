#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_fixed_base_gpu.py -x -q -m gpu 2>&1 | tail -8
