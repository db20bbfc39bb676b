# usage: prof_entry.sh <what> [reps]  -> per-kernel averages (rocprofv3 --kernel-trace --stats) of tools/run_entry.py <what>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; W=$1; N=${2:-6}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pe_$W --output-format csv -- python3 $R/tools/run_entry.py $W $N > /dev/null 2>&1
cp $R/gpurun_out/pe_$W/*/*kernel_stats.csv $R/gpurun_out/r2_kstats_$W.csv; rm -rf $R/gpurun_out/pe_$W
