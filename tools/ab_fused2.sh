run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"; }
run OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=16
run OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=8
run OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=8 OZK_MSM_FIN_MAX=1
run OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=8 OZK_MSM_FIN_MAX=2
run OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=4
run OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=8 OZK_MSM_WSUM0_IN_TAIL=1 OZK_MSM_WSUM0_PRIO=0
run OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=8 OZK_MSM_WSUM0_IN_TAIL=1 OZK_MSM_WSUM0_PRIO=1
