#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3_09; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_pipeline3_gpu.py -x -q 2>&1 | tail -3 | tee -a $O/summary.txt
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
import ctypes
from octopuszk_amd import lib
L = lib.load()
lib.check(L.ozk_prof_enable(2)); print("device clock: %.1f kHz" % L.ozk_prof_clock_khz()); lib.check(L.ozk_prof_enable(0))
PY
python tools/sched_probe.py --sched p3 --depth 4 --reps 200 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/sched_probe.py --sched p3 --depth 4 --reps 200 --prof 2 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace_p3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/sched_probe.py --sched p3 --depth 4 --reps 30 > $O/prof_p3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline_steady.py $O/trace_p3 2 3 > $O/timeline_p3.txt 2>&1
rm -rf $O/trace_p3
cat $O/timeline_p3.txt
