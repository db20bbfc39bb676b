#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_45; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace -d $O/tr --output-format csv -- python3 $R/tools/host_path.py 20 > $O/log.txt 2>&1
grep "double" $O/log.txt | cut -c1-200
ls $O/tr/*/ | head
cd $R
python3 - <<'PY'
import csv, glob, os
d = glob.glob(os.environ.get('GRAFT_REPO_ROOT','.') + '/gpurun_out/r3_45/tr/*/')[0]
kt = list(csv.DictReader(open(glob.glob(d + '*kernel_trace.csv')[0])))
mc = list(csv.DictReader(open(glob.glob(d + '*memory_copy_trace.csv')[0])))
print(mc[0].keys())
ev = []
for r in kt: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K q%s %s' % (r['Queue_Id'], r['Kernel_Name'].replace('void ozk::','').replace('ozk::','')[:40])))
for r in mc: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s' % ' '.join('%s=%s' % (k, r[k]) for k in r if k in ('Direction','Size','Bytes','Stream_Id','Kind','Src_Agent_Id','Dst_Agent_Id'))))
ev.sort()
# find G2 level-1 kernels at 2^18 of the double MSM: k_segreduce<G2Cfg, true
g2 = [i for i, e in enumerate(ev) if 'k_segreduce<G2Cfg, true' in e[2]]
print('G2 level-1 launches:', len(g2))
# the double MSM calls are the first 5 of them (then fixed-base G2 etc. do not use segreduce)
lo = max(0, g2[0] - 60); hi = min(len(ev), g2[4] + 40) if len(g2) >= 5 else len(ev)
t0 = ev[lo][0]; prev_end = t0
with open(os.environ.get('GRAFT_REPO_ROOT','.') + '/gpurun_out/r3_45/double_timeline.txt', 'w') as f:
    for s, e, name in ev[lo:hi]:
        gap = (s - prev_end) / 1e3
        f.write('%10.1f .. %10.1f %9.1f us %s%s\n' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, name, '   <== idle %.0f us before' % gap if gap > 500 else ''))
        prev_end = max(prev_end, e)
PY
rm -rf $O/tr
grep -n "idle" $O/double_timeline.txt | head -20
