cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/p3 --output-format csv -- python3 $R/tools/pipe3.py 12 > $R/gpurun_out/r2_pipe3_prof.log 2>&1
cd $R && python - <<'PY'
import csv, glob
t = glob.glob('gpurun_out/p3/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
# the three-stage phase: before the first k_wsum_fused of ... take dispatches 40%..60% of the three-stage part
idx = [i for i, r in enumerate(rows) if 'k_segreduce<ozk::G1Cfg, true>' in r['Kernel_Name']]
# three-stage part = first 18 level-1 launches (6 warm-up + 12)
lo, hi = idx[9], idx[12]
t0 = int(rows[lo]['Start_Timestamp'])
qs = {}
with open('gpurun_out/r2_timeline_pipe3.txt', 'w') as f:
    for r in rows[lo - 20:hi + 1]:
        q = qs.setdefault(r['Queue_Id'], len(qs))
        f.write("q%d %9.1f .. %9.1f %8.1f us  %s\n" % (q, (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3,
                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'].replace('void ozk::', '').replace('ozk::', '')[:50]))
PY
rm -rf gpurun_out/p3
