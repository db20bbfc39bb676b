"""Per-kernel averages of rocprofv3 --pmc counters (usage: pmc_summary.py <dir> [<dir> ...]).
Each directory is one `rocprofv3 --pmc <COUNTER...> --kernel-trace` pass; prints CSV
kernel,launches,<COUNTER>_avg,... and, with --traffic-json PATH, rewrites the level-1 kernel's
HBM bytes per launch (FETCH_SIZE / WRITE_SIZE are reported in KB; gfx950 correction: see
/opt/skills/guides/MI355X_MICROARCH.md and profiles/traffic.json)."""
import collections
import csv
import glob
import json
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
tj = None
if "--traffic-json" in sys.argv:
    tj = sys.argv[sys.argv.index("--traffic-json") + 1]
    args.remove(tj)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(lambda: collections.defaultdict(float))
        names = {}
        for r in csv.DictReader(open(f)):
            key = r["Dispatch_Id"]
            names[key] = r["Kernel_Name"]
            per_dispatch[key][r["Counter_Name"]] += float(r["Counter_Value"])
        for key, cs in per_dispatch.items():
            for c, v in cs.items():
                acc[names[key]][c].append(v)
counters = sorted({c for k in acc for c in acc[k]})
print("kernel,launches," + ",".join(c + "_avg" for c in counters))
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    print('"%s",%d,%s' % (k, n, ",".join("%.1f" % (sum(acc[k][c]) / len(acc[k][c])) if acc[k][c] else "" for c in counters)))
if tj:
    k = [x for x in acc if "k_segreduce" in x and "true" in x and "G1" in x][0]
    fetch = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
    write = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
    j = json.load(open(tj))
    j["FETCH_SIZE_KB"] = fetch
    j["WRITE_SIZE_KB"] = write
    j["k_segreduce_level1_hbm_bytes_per_launch"] = int((2 * fetch + write) * 1024)
    j["k_segreduce_level1_hbm_bytes_per_launch_uncorrected"] = int((fetch + write) * 1024)
    json.dump(j, open(tj, "w"), indent=1)
