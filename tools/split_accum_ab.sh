#!/bin/bash
# Two questions about the streams of the three-stage schedule, same box, interleaved (tools/sched_probe.py, 200 MSMs):
#   split : the accumulate stage in two parts (OZK_P3_SPLIT_ACCUM=1: level 1 alone on its stream, the rest — run merge,
#           generic levels — at the head of the tail stream or on a stream of its own), with and without CU masks
#   queues: HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and the shipped schedule
#           uses exactly four streams (null / sort, accumulate, two tails): what a fifth stream costs, and what eight
#           queues change
#   bash tools/split_accum_ab.sh split|queues > gpurun_out/<...>.txt
set -o pipefail
run() { timeout -k 10 120 python tools/sched_probe.py --reps 200 --depth 4 --prof 2 "$@" 2>&1 | tail -1; }
M="--own-sort-stream --acc-mask comp"
case "${1:-split}" in
split)
  for rnd in 1 2 3; do
    OZK_P3_SPLIT_ACCUM=0 run
    echo -n "SPLIT "; OZK_P3_SPLIT_ACCUM=1 run
    echo -n "SPLIT rest=own "; OZK_P3_SPLIT_ACCUM=1 run --rest-stream own
    echo -n "MASK "; OZK_P3_SPLIT_ACCUM=0 run $M --tail-cus 32
    echo -n "SPLIT+MASK "; OZK_P3_SPLIT_ACCUM=1 run $M --tail-cus 32
    echo -n "SPLIT+MASK rest=own "; OZK_P3_SPLIT_ACCUM=1 run $M --tail-cus 32 --rest-stream own
    echo -n "SPLIT+MASK rest=comp "; OZK_P3_SPLIT_ACCUM=1 run $M --tail-cus 32 --rest-stream comp
  done ;;
queues)
  for rnd in 1 2; do
    for q in 4 8; do
      export GPU_MAX_HW_QUEUES=$q
      echo -n "Q=$q "; OZK_P3_SPLIT_ACCUM=0 run
      echo -n "Q=$q own-sort "; OZK_P3_SPLIT_ACCUM=0 run --own-sort-stream
      echo -n "Q=$q ts3 "; OZK_P3_SPLIT_ACCUM=0 run --tail-streams 3 --depth 6
      echo -n "Q=$q SPLIT "; OZK_P3_SPLIT_ACCUM=1 run
      echo -n "Q=$q SPLIT rest=own "; OZK_P3_SPLIT_ACCUM=1 run --rest-stream own
      echo -n "Q=$q SPLIT rest=own ts3 "; OZK_P3_SPLIT_ACCUM=1 run --rest-stream own --tail-streams 3 --depth 6
    done
  done ;;
esac
