cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ln in 16 17 18; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ps$ln --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-streams-leg --steps 6 --in-flight 1 --logn $ln > /dev/null 2>&1
(cd $R && echo "== 2^$ln" && python tools/prof_summary.py gpurun_out/ps$ln ; rm -rf gpurun_out/ps$ln)
done
