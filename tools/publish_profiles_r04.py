#!/usr/bin/env python3
"""Copy the evidence of tools/collect_profiles_r04.sh (gpurun_out/evidence4/r04_*) into profiles/ and refresh
profiles/traffic.json from the PMC summary of the same run (FETCH_SIZE / WRITE_SIZE are in KB; gfx950: reads x2,
/opt/skills/guides/MI355X_MICROARCH.md, HBM section).  Also writes profiles/r04_kernel_resources.txt from the local
build (tools/kernel_resources.py).  usage: publish_profiles_r04.py"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(ROOT, "gpurun_out", "evidence4")
P = os.path.join(ROOT, "profiles")
man = open(os.path.join(E, "r04_MANIFEST.txt")).read().split("\n")
commit = man[0].split()[-1]
for f in sorted(os.listdir(E)):
    if f.startswith("r04_"):
        shutil.copy(os.path.join(E, f), os.path.join(P, f))
rows = list(csv.DictReader(open(os.path.join(E, "r04_pmc_hbm_summary.csv"))))
k = [r for r in rows if "k_segreduce" in r["kernel"] and "G1Cfg, true" in r["kernel"]][0]
fetch, write = float(k["FETCH_SIZE_avg"]), float(k["WRITE_SIZE_avg"])
tj = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --no-cpu-baseline --timed-only "
                "--steps 4 --warmup 1`; per-launch averages over %s launches, profiles/r04_pmc_hbm_summary.csv" % k["launches"],
      "kernel": k["kernel"], "collected_on_commit": commit, "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
      "k_segreduce_level1_hbm_bytes_per_launch": int((2 * fetch + write) * 1024),
      "k_segreduce_level1_hbm_bytes_per_launch_uncorrected": int((fetch + write) * 1024),
      "note": "gfx950 correction: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM counters); algorithmic bytes per launch: 134217728"}
json.dump(tj, open(os.path.join(P, "traffic.json"), "w"), indent=1)
res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py")], capture_output=True, text=True)
open(os.path.join(P, "r04_kernel_resources.txt"), "w").write("# commit %s (local build)\n" % commit + res.stdout)
print("published %s; level-1 HBM bytes per launch %d" % (commit, tj["k_segreduce_level1_hbm_bytes_per_launch"]))
