run() { echo "== $*"; env "$@" timeout -k 10 100 python bench.py --no-cpu-baseline --schedule streams 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"; }
run A=1
run OZK_MSM_L1=32
run OZK_MSM_L1=32 OZK_L1_LDS=0
run OZK_MSM_L1=36
run OZK_MSM_L1=28
echo "== streams 4"; timeout -k 10 100 python bench.py --no-cpu-baseline --schedule streams --in-flight 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"
echo "== streams 2"; timeout -k 10 100 python bench.py --no-cpu-baseline --schedule streams --in-flight 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"
echo "== pipeline L1=32 nolds"; OZK_MSM_L1=32 OZK_L1_LDS=0 timeout -k 10 100 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"
echo "== pipeline 3 L1=32"; OZK_MSM_L1=32 timeout -k 10 100 python bench.py --no-cpu-baseline --in-flight 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['single_msm_latency_ms'], d['roofline']['kernel_avg_ms'])"
