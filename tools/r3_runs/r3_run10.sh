#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_10; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
