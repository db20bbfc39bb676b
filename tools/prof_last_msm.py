import csv, glob, sys
d = sys.argv[1]
t = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(t)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_finalize' in r['Kernel_Name']]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0=int(rows[lo]['Start_Timestamp']); tot=0
for r in rows[lo:hi]:
    dur=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot+=dur
    if dur>12: print("%-50s %9.1f us start=%8.1f" % (r['Kernel_Name'].replace('void ozk::','').replace('ozk::','')[:50], dur, (int(r['Start_Timestamp'])-t0)/1e3))
print("sum %.1f"%tot)
