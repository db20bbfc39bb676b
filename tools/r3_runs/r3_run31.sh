#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3_31; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/ktb --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 5 > $O/ktb.log 2>&1
rocprofv3 --kernel-trace -d $O/ktp --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/sched_probe.py --sched p3 --depth 4 --reps 20 --prof 2 > $O/ktp.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline_range.py $O/ktb 20 25 > $O/timeline_bench.txt
python tools/timeline_range.py $O/ktp 20 25 > $O/timeline_probe.txt
rm -rf $O/ktb $O/ktp
grep -o '"ms_per_step": [0-9.]*' $O/ktb.log; grep "per MSM" $O/ktp.log
tail -1 $O/timeline_bench.txt; tail -1 $O/timeline_probe.txt
