#!/bin/bash
# A/B the MSM tuning knobs (environment variables read by make_plan / the driver, msm_var.hip):
#   tools/ab_env.sh "OZK_MSM_L1=32" "OZK_MSM_L1=40 OZK_MSM_S=8" ...
# prints value (Mscalar-mul/s, two in flight), ms/step, single-MSM latency, level-1 kernel ms.
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python bench.py --no-cpu-baseline --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"
done
