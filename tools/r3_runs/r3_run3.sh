#!/bin/bash
# schedule knob sweep under the three-stage schedule (each line: knob, throughput, level-1 kernel ms)
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3_03; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() { echo -n "$1 | " | tee -a $O/summary.txt; env $1 python tools/sched_probe.py --sched p3 --depth 4 --tail-streams 2 --prof ${2} 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt; }
run "A=0"
run "A=0" "--depth 2 --tail-streams 1"
python tools/sched_probe.py --sched p3 --depth 4 --tail-streams 2 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
python tools/sched_probe.py --sched p2 --prof 2>&1 | grep -v amdgpu.ids | tee -a $O/summary.txt
for S in 8 16 32; do run "OZK_MSM_S=$S"; done
run "OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=8"
run "OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=16"
run "OZK_MSM_TAIL_SERIAL_ABOVE=64"
run "OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=64"
run "OZK_MSM_FIN_MAX=16"
for L in 40 48 64; do run "OZK_MSM_L1_MIN=$L"; done
run "OZK_L1_LDS=81920"
run "OZK_L1_LDS=0"
run "OZK_MSM_WSUM0_PRIO=1"
run "A=1"
for sched in pipeline3 pipeline; do
  python bench.py --steps 20 --warmup 5 --schedule $sched --no-cpu-baseline --no-streams-leg 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $sched 20 steps: value=%.1f ms/step=%.4f k_ms=%.4f single=%s' % (j['value'], j['ms_per_step'], j['roofline']['kernel_avg_ms'], j['config']['single_msm_latency_ms']))" | tee -a $O/summary.txt
done
