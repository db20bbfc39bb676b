#!/bin/bash
# same-box A/B of the twiddle pyramid (round 4): per-pass kernel times of the 2^22 transform and the 2^21 witness map
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do
  export OZK_FFT_TW_PYRAMID=$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt_fft --output-format csv -- python3 $R/tools/run_entry.py fft22 20 > /dev/null 2>&1
  echo "== OZK_FFT_TW_PYRAMID=$v"; python3 -c "
import csv,glob
t=0
for r in csv.DictReader(open(glob.glob('$R/gpurun_out/kt_fft/*/*kernel_stats.csv')[0])):
    if 'k_fft_pass' in r['Name']:
        print('  %-60s avg %.1f us' % (r['Name'][:60], float(r['AverageNs'])/1e3)); t+=float(r['AverageNs'])/1e3
print('  sum %.1f us' % t)"
  rm -rf $R/gpurun_out/kt_fft
  python3 $R/tools/run_entry.py qap21 20 2>&1 | tail -1
done
