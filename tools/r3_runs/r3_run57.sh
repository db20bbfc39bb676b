#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_fft1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f --output-format csv -- python3 $R/tools/run_entry.py fft22 3 > $O/f.log 2>&1 || { echo "fetch pass failed"; tail -5 $O/f.log; exit 1; }
echo "fetch done"
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w --output-format csv -- python3 $R/tools/run_entry.py fft22 3 > $O/w.log 2>&1 || { echo "write pass failed"; exit 1; }
echo "write done"
cd $R && python3 tools/pmc_summary.py $O/f $O/w 2>&1 | grep "fft_pass\|kernel," | cut -c1-300
rm -rf $O
