#!/usr/bin/env python3
"""Fill and drain of a burst of pipelined MSMs: from a rocprofv3 --kernel-trace of `bench.py --timed-only --steps K
--warmup W`, the level-1 launches of the LAST K MSMs (start, duration, gap since the previous one ended), the start of
the burst's first kernel and the end of its last.   usage: burst_timeline.py <trace dir> <K>"""
import csv, glob, sys
d, K = sys.argv[1], int(sys.argv[2])
t = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
ev = [(r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
l1 = [e for e in ev if 'k_segreduce' in e[0] and 'true, true' in e[0]]
fin = [e for e in ev if 'k_finalize' in e[0]]
dig = [e for e in ev if 'k_digits_glv' in e[0]]
l1, fin, dig = l1[-K:], fin[-K:], dig[-K:]
t0 = min(e[1] for e in dig)
# the memset in front of the first digits kernel belongs to the burst too
first = max([e for e in ev if e[2] <= dig[0][1] and 'fillBuffer' in e[0]], key=lambda e: e[1], default=dig[0])
t0 = min(t0, first[1])
tend = max(e[2] for e in fin)
print("burst of %d MSMs: %.3f ms from the first kernel's start to the last finalize's end = %.3f ms per MSM" % (K, (tend - t0) / 1e6, (tend - t0) / 1e6 / K))
prev = None
for i, e in enumerate(l1):
    gap = (e[1] - prev) / 1e3 if prev else (e[1] - t0) / 1e3
    print("  level 1 #%2d: starts %8.1f us, runs %7.1f us, %s %6.1f us" % (i, (e[1] - t0) / 1e3, (e[2] - e[1]) / 1e3, "gap since previous" if prev else "after burst start", gap))
    prev = e[2]
print("  after the last level 1: %.1f us until the last finalize ends" % ((tend - prev) / 1e3))
for i, e in enumerate(fin[-3:]):
    print("  finalize #%d ends at %.1f us" % (K - 3 + i, (e[2] - t0) / 1e3))
