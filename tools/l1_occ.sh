run() { echo "== $*"; env "$@" timeout -k 10 100 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"; }
run A=1
run OZK_L1_LDS=0
run OZK_L1_LDS=0 OZK_MSM_L1=32
run OZK_MSM_L1=32
run OZK_L1_LDS=0 OZK_MSM_L1=48
echo "== streams"; timeout -k 10 100 python bench.py --no-cpu-baseline --schedule streams 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"
echo "== streams nolds"; OZK_L1_LDS=0 timeout -k 10 100 python bench.py --no-cpu-baseline --schedule streams 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"
