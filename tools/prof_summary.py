"""Print the per-dispatch timeline of the last bench step from a rocprofv3 kernel trace CSV."""
import csv, glob, sys
d = sys.argv[1]
t = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(t)))
names = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Grid_Size_X'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
idx = [i for i, n in enumerate(names) if 'k_finalize' in n[0]]
lo, hi = idx[-2] + 1, idx[-1] + 1
t0 = names[lo][3]
tot = 0
for n in names[lo:hi]:
    short = n[0].replace('void ozk::', '').replace('ozk::', '')[:46]
    print("%-48s %9.1f us grid=%-9s start=%8.1f" % (short, n[1], n[2], (n[3] - t0) / 1e3))
    tot += n[1]
print("sum of kernel time %.1f us; span %.1f us" % (tot, (names[hi - 1][4] - t0) / 1e3))
