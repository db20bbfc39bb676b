"""Per-kernel average durations and one steady-state step's two-stream timeline from a rocprofv3
kernel trace of `bench.py --in-flight 2` (usage: prof_pipeline.py <dir>)."""
import csv, glob, sys, collections
d = sys.argv[1]
t = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n): return n.replace('void ozk::', '').replace('ozk::', '')[:40]
fin = [i for i, r in enumerate(rows) if 'k_finalize' in r['Kernel_Name']]
# steady state: between finalize #10 and #12
lo, hi = fin[9], fin[11]
t0 = int(rows[lo]['Start_Timestamp'])
agg = collections.defaultdict(list)
for r in rows[fin[5]:fin[-6]]:
    agg[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("average duration while two MSMs are in flight (us):")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("  %-42s n=%4d avg=%8.1f  total/step=%8.1f" % (k, len(v), sum(v) / len(v), sum(v) / (len(fin) - 11)))
print("timeline of two consecutive steps (queue id, start, duration):")
for r in rows[lo:hi + 1]:
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if dur < 15: continue
    print("  q%-3s %-42s start=%8.1f dur=%8.1f" % (r.get('Queue_Id', '?'), short(r['Kernel_Name']), (int(r['Start_Timestamp']) - t0) / 1e3, dur))
