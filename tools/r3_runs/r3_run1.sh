#!/bin/bash
# round 3, call 1: parity of the reshaped sort + first look at the three-stage schedule
set -o pipefail
O=gpurun_out/r3_01; mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
timeout -k 10 120 python tools/pipe3.py 100 > $O/pipe3.log 2>&1; cat $O/pipe3.log | tee -a $O/summary.txt
for sched in pipeline3 pipeline; do
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --schedule $sched --no-cpu-baseline --no-streams-leg > $O/bench_${sched}_$rep.json 2> $O/bench_${sched}_$rep.err
    python - <<PY | tee -a $O/summary.txt
import json
try:
    j = json.loads(open("$O/bench_${sched}_$rep.json").read().strip().splitlines()[-1])
    print("$sched rep$rep value=%.1f ms/step=%.4f k_ms=%.4f single=%s" % (j["value"], j["ms_per_step"], j["roofline"]["kernel_avg_ms"], j["config"]["single_msm_latency_ms"]))
except Exception as e:
    print("$sched rep$rep FAILED", e)
PY
  done
done
timeout -k 10 200 python bench.py --steps 100 --warmup 5 --schedule pipeline3 --no-cpu-baseline --no-streams-leg > $O/bench_p3_100.json 2> $O/bench_p3_100.err
python -c "import json;j=json.loads(open('$O/bench_p3_100.json').read().strip().splitlines()[-1]);print('p3 100 steps value=%.1f ms/step=%.4f k_ms=%.4f'%(j['value'],j['ms_per_step'],j['roofline']['kernel_avg_ms']))" | tee -a $O/summary.txt
OZK_BENCH_TS=1 timeout -k 10 200 python bench.py --steps 100 --warmup 5 --schedule pipeline3 --tail-streams 1 --no-cpu-baseline --no-streams-leg > $O/bench_p3_ts1.json 2> $O/bench_p3_ts1.err
python -c "import json;j=json.loads(open('$O/bench_p3_ts1.json').read().strip().splitlines()[-1]);print('p3 ts1 100 steps value=%.1f ms/step=%.4f k_ms=%.4f'%(j['value'],j['ms_per_step'],j['roofline']['kernel_avg_ms']))" | tee -a $O/summary.txt
OZK_L1_LDS=49152 timeout -k 10 200 python bench.py --steps 100 --warmup 5 --schedule pipeline3 --no-cpu-baseline --no-streams-leg > $O/bench_p3_lds48.json 2> $O/bench_p3_lds48.err
python -c "import json;j=json.loads(open('$O/bench_p3_lds48.json').read().strip().splitlines()[-1]);print('p3 L1_LDS=48K value=%.1f ms/step=%.4f k_ms=%.4f'%(j['value'],j['ms_per_step'],j['roofline']['kernel_avg_ms']))" | tee -a $O/summary.txt
exit 0
