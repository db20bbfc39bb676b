#!/bin/bash
# Does confining the latency-bound TAIL kernels (window sums, Horner) to a few CUs leave the bucket accumulation more of
# the chip?  Same box, interleaved.   bash tools/cumask_ab.sh probe|family|bench > gpurun_out/cumask_<section>.txt
#   probe : the three-stage schedule (200 MSMs of 2^20, tools/sched_probe.py) with the tail streams created through
#           hipExtStreamCreateWithCUMask, the accumulate stream on everything or on the complement; --prof 2 = level-1
#           times from the kernel's own clock stamps.  The sort stage runs on a stream of its own: a CU-masked stream is
#           a BLOCKING stream and serialises against the null stream (3.0-3.5 ms per MSM when the sort stays there).
#   family: the shipped schedule (sort on the null stream, no masks) against the best masked family
#   bench : bench.py in the driver's form and over 100 steps with and without --tail-cus 32
set -o pipefail
run() { timeout -k 10 120 python tools/sched_probe.py --reps 200 --depth 4 --prof 2 "$@" 2>&1 | tail -1; }
case "${1:-family}" in
probe)
  for rnd in 1 2; do
    run --own-sort-stream
    for n in 16 32 64 96; do
      run --own-sort-stream --tail-cus $n
      run --own-sort-stream --tail-cus $n --acc-mask comp
    done
    run --own-sort-stream --tail-cus 64 --sort-mask tail --acc-mask comp
    run --own-sort-stream --tail-cus 64 --sort-mask comp --acc-mask comp
    run --own-sort-stream --tail-cus 32 --sort-mask comp --acc-mask comp
    run --own-sort-stream --tail-cus 32 --mask-layout high
  done ;;
family)
  for rnd in 1 2 3; do
    run
    for n in 24 32 40 48; do run --own-sort-stream --tail-cus $n --acc-mask comp; done
    run --class-tail-cus 32
    run --own-sort-stream --tail-cus 32 --acc-mask comp --tail-streams 1
    run --own-sort-stream --tail-cus 32 --acc-mask comp --tail-streams 3 --depth 6
  done ;;
bench)
  for rnd in 1 2; do
    for tc in 0 32; do
      for st in "--steps 20 --warmup 5" "--steps 100 --warmup 5"; do
        timeout -k 10 300 python bench.py --gpus 1 $st --no-cpu-baseline --tail-cus $tc 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
r = j['roofline']
print('tail_cus=%-3s steps=%-4d value %.1f  ms/step %.3f  level-1 in-schedule %.3f ms alone %s  clock %s' % ('$tc', j['steps'], j['value'], j['ms_per_step'], r['kernel_avg_ms'], (r.get('kernel_ms_alone') or {}).get('median'), r.get('shader_clock_mhz', {}).get('in_schedule', {}).get('median')))"
      done
    done
  done ;;
esac
