#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3_02; mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/pipe3.py 100 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/sched_probe.py --sched p3 --depth 2 --tail-streams 1 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/sched_probe.py --sched p3 --depth 2 --tail-streams 1 --own-sort-stream 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/sched_probe.py --sched p3 --depth 2 --tail-streams 1 --prof 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/sched_probe.py --sched p3 --depth 4 --tail-streams 2 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/sched_probe.py --sched p3 --depth 4 --tail-streams 2 --prof 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/sched_probe.py --sched p2 --prof 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace_p3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/sched_probe.py --sched p3 --depth 2 --tail-streams 1 --reps 24 > $O/prof_p3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline_steady.py $O/trace_p3 2 3 > $O/timeline_p3.txt 2>&1
rm -rf $O/trace_p3
tail -40 $O/timeline_p3.txt
