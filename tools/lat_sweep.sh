run() { echo "== $*"; env "$@" timeout -k 10 100 python bench.py --no-cpu-baseline --in-flight 1 --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"; }
run A=1
run OZK_MSM_TAIL_SERIAL_ABOVE=64
run OZK_MSM_TAIL_SERIAL_ABOVE=16
run OZK_MSM_S=8
run OZK_MSM_S=2
run OZK_MSM_S=8 OZK_MSM_TAIL_SERIAL_ABOVE=64
run OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=8
run OZK_MSM_WSUM_FUSED=0 OZK_MSM_S=16
run OZK_MSM_FIN_MAX=8
run OZK_MSM_FIN_MAX=2
run OZK_L1_LDS=0
run OZK_MSM_L1=43
