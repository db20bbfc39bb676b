#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_sharded_gpu.py -x -q -m gpu -k "in_process_sharded_entry_equals_single_call" 2>&1 | tail -40
