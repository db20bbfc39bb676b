#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_25; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/kt --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 5 > $O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline_range.py $O/kt 20 25 > $O/timeline20.txt
rm -rf $O/kt
head -60 $O/timeline20.txt; echo ...; tail -45 $O/timeline20.txt
