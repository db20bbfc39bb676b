#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "
import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 900 python -m pytest tests/test_fixed_base_gpu.py tests/test_var_msm_gpu.py -x -q -m gpu 2>&1 | tail -2
