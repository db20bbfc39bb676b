#!/usr/bin/env python3
"""Copy the evidence of tools/collect_profiles_r02.sh (gpurun_out/evidence2/) into profiles/ under round-2 names
and refresh profiles/traffic.json from the PMC summary (FETCH_SIZE / WRITE_SIZE are in KB; gfx950: reads x2,
/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  usage: publish_profiles_r02.py <commit>"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(ROOT, "gpurun_out", "evidence2")
P = os.path.join(ROOT, "profiles")
commit = sys.argv[1]
for f in os.listdir(E):
    src = os.path.join(E, f)
    if not os.path.isfile(src) or f.endswith((".err", ".log")):
        continue
    if f.startswith("bench_"):
        line = [l for l in open(src) if l.startswith("{")]
        if line:
            open(os.path.join(P, "r02_" + f), "w").write(line[-1])
        continue
    name = f.replace("timeline_single.txt", "timeline_single_msm.txt")
    shutil.copy(src, os.path.join(P, "r02_" + name))
rows = list(csv.DictReader(open(os.path.join(E, "pmc_hbm_summary.csv"))))
k = [r for r in rows if "k_segreduce" in r["kernel"] and "true" in r["kernel"] and "G1" in r["kernel"]][0]
fetch, write = float(k["FETCH_SIZE_avg"]), float(k["WRITE_SIZE_avg"])
tj = json.load(open(os.path.join(P, "traffic.json")))
tj.update({"collected_on_commit": commit, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
           "k_segreduce_level1_hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
           "k_segreduce_level1_hbm_bytes_per_launch_uncorrected": int((fetch + write) * 1024)})
tj["source"] = tj["source"].split("; per-launch")[0] + "; per-launch averages over %s launches, profiles/r02_pmc_hbm_summary.csv" % k["launches"]
json.dump(tj, open(os.path.join(P, "traffic.json"), "w"), indent=1)
print("published; traffic", tj["k_segreduce_level1_hbm_bytes_per_launch"])
