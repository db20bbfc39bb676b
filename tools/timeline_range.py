#!/usr/bin/env python3
"""Every dispatch of at least MIN_US microseconds in a rocprofv3 kernel trace, relative to the start of the LAST run of
N consecutive level-1 launches (the timed region of `bench.py --timed-only --steps N`), with its hardware queue.
usage: timeline_range.py DIR N [min_us=20]"""
import csv, glob, sys
d, N = sys.argv[1], int(sys.argv[2])
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
t = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
l1 = [i for i, r in enumerate(rows) if 'k_segreduce<ozk::G1Cfg, true' in r['Kernel_Name']]
first = l1[-N]
# the sort of the first timed MSM precedes its level-1 launch: back up to the digits kernel before it
lo = first
while lo > 0 and 'k_digits' not in rows[lo]['Kernel_Name']:
    lo -= 1
t0 = int(rows[lo]['Start_Timestamp'])
qs = {}
for r in rows[lo:]:
    q = qs.setdefault(r['Queue_Id'], len(qs))
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if (e - s) / 1e3 < min_us:
        continue
    print("q%d %9.1f .. %9.1f %8.1f us  %s" % (q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3,
          r['Kernel_Name'].replace('void ozk::', '').replace('ozk::', '')[:44]))
print("span %.1f us" % ((max(int(r['End_Timestamp']) for r in rows[lo:]) - t0) / 1e3))
