P='import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["config"]["single_msm_latency_ms"], d["roofline"]["kernel_avg_ms"], d["config"].get("three_streams_Mscalar_mul_s"))'
echo "pipeline default"; python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P"
echo "streams"; python bench.py --no-cpu-baseline --schedule streams 2>/dev/null | python -c "$P"
echo "pipeline HWQ=8"; GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline 2>/dev/null | python -c "$P"
echo "streams HWQ=8"; GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --schedule streams 2>/dev/null | python -c "$P"
