cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pf_big --output-format csv -- python3 $R/tools/run_entry.py fft22 10 > /dev/null 2>&1
cp $R/gpurun_out/pf_big/*/*kernel_stats.csv $R/gpurun_out/r2_fft_big_stats.csv; rm -rf $R/gpurun_out/pf_big
