for rep in 1 2; do
for cfg in "--schedule pipeline --in-flight 2" "--schedule pipeline --in-flight 3" "--schedule streams --in-flight 2" "--schedule streams --in-flight 3" "--schedule streams --in-flight 4"; do
  echo -n "$cfg: "
  timeout -k 10 120 python bench.py --no-cpu-baseline --steps 60 $cfg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_avg_ms'])"
done; done
