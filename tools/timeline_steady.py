#!/usr/bin/env python3
"""Steady-state timeline from a rocprofv3 kernel trace: every dispatch between the level-1 launch number M and number
M + K (default: the middle of the trace, K = 2), with the hardware queue and the overlap with level-1 kernels,
then per-kernel averages over that window.   usage: timeline_steady.py DIR [K] [min_us]"""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
t = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
l1 = [i for i, r in enumerate(rows) if 'k_segreduce<ozk::G1Cfg, true' in r['Kernel_Name']]
m = len(l1) // 2
lo, hi = l1[m], l1[m + K]
t0 = int(rows[lo]['Start_Timestamp'])
t1 = int(rows[hi]['Start_Timestamp'])
qs = {}
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if e < t0 or s > t1:
        continue
    q = qs.setdefault(r['Queue_Id'], len(qs))
    dur = (e - s) / 1e3
    if dur < min_us:
        continue
    print("q%d %9.1f .. %9.1f %8.1f us  %s" % (q, (s - t0) / 1e3, (e - t0) / 1e3, dur,
          r['Kernel_Name'].replace('void ozk::', '').replace('ozk::', '')[:48]))
print("period: %.1f us per MSM over %d MSMs" % ((t1 - t0) / 1e3 / K, K))
# averages over the middle half of the trace
a, b = int(rows[l1[len(l1) // 4]]['Start_Timestamp']), int(rows[l1[3 * len(l1) // 4]]['Start_Timestamp'])
tot, cnt = defaultdict(float), defaultdict(int)
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s < a or s >= b:
        continue
    nm = r['Kernel_Name'].replace('void ozk::', '').replace('ozk::', '')[:40]
    tot[nm] += (e - s) / 1e3
    cnt[nm] += 1
print("per-kernel averages (us) over the middle half, and per MSM:")
for nm in sorted(tot, key=lambda k: -tot[k]):
    print("  %-42s n=%4d avg=%8.1f" % (nm, cnt[nm], tot[nm] / cnt[nm]))
