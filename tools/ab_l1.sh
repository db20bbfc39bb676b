run() { echo "== $*"; env "$@" timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"; }
run OZK_MSM_L1=32
run OZK_MSM_L1=24
run OZK_MSM_L1=40
run OZK_MSM_L1=48
run OZK_MSM_L1=64
run OZK_MSM_L1=32 OZK_MSM_S=8
