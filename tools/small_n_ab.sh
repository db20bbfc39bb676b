#!/bin/bash
# same-box A/B of the small-MSM plan rules (round 4): sizes 2^10 .. 2^16
for kn in "" "OZK_MSM_SMALL_SORT=0" "OZK_MSM_L1=40" "OZK_MSM_L1=40 OZK_MSM_SMALL_SORT=0 OZK_MSM_S_LAT=8" "OZK_MSM_L1=16" "OZK_MSM_S_LAT=8" "OZK_MSM_S_LAT=4"; do
  echo "== $kn"; env $kn python tools/size_sweep.py 10 16 2>&1 | grep "n=" | awk '{printf "%s %s  ", $2, $(NF-1)} END {print ""}'
done
