"""Per-dispatch timeline of the LAST occurrence of a workload in a rocprofv3 kernel trace: from the last dispatch
whose name contains START (argv[2]) to the end of the trace, with the queue each dispatch ran on.
usage: prof_timeline.py DIR START_SUBSTRING [min_us]"""
import csv, glob, sys
d, start = sys.argv[1], sys.argv[2]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
t = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if start in r['Kernel_Name']]
# first dispatch of the last burst of START kernels
lo = idx[-1]
while lo - 1 in idx:
    lo -= 1
t0 = int(rows[lo]['Start_Timestamp'])
qs = {}
for r in rows[lo:]:
    q = r.get('Queue_Id', '?')
    qs.setdefault(q, len(qs))
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if dur < min_us:
        continue
    short = r['Kernel_Name'].replace('void ozk::', '').replace('ozk::', '')[:60]
    print("q%d %9.1f .. %9.1f  %8.1f us  grid=%-9s %s" % (qs[q], (int(r['Start_Timestamp']) - t0) / 1e3,
          (int(r['End_Timestamp']) - t0) / 1e3, dur, r['Grid_Size_X'], short))
print("span %.1f us" % ((max(int(r['End_Timestamp']) for r in rows[lo:]) - t0) / 1e3))
