# per-kernel timeline of a lone 2^20 G1 MSM (bench.py --in-flight 1 --timed-only under rocprofv3) -> gpurun_out/r2_timeline_single.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ps --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 10 --in-flight 1 > /dev/null 2>&1
cd $R && python tools/prof_summary.py gpurun_out/ps > gpurun_out/r2_timeline_single.txt; rm -rf gpurun_out/ps
