"""Print name, calls, average us of the kernels matching a substring from a rocprofv3 --stats directory:
   stats_grep.py <dir> <substring> [<substring> ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print("%-70s calls=%-4s avg=%9.1f us" % (r["Name"].replace("void ozk::", "").replace("unsigned int", "u32")[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
