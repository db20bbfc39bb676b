#!/bin/bash
# Collect the judged evidence on the GPU box (run through gpurun from the repo root):
#   bench lines, rocprofv3 kernel stats of the same commands, PMC HBM counters (separate passes),
#   per-kernel timelines.  Outputs under gpurun_out/evidence/; copy what should be judged to profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
rm -rf $O && mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --in-flight 1 --no-cpu-baseline > $O/bench_in_flight_1.json 2> $O/bench_in_flight_1.err
python tools/perf_all.py > $O/perf_all.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt_default --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 20 > $O/kt_default.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_single --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 10 --in-flight 1 > $O/kt_single.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --in-flight 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --in-flight 1 > $O/pmc_write.log 2>&1
cd $R
python tools/prof_summary.py $O/kt_single > $O/timeline_single.txt
python tools/prof_pipeline.py $O/kt_default > $O/timeline_pipelined.txt
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_hbm_summary.csv
cp $O/kt_default/*/*kernel_stats.csv $O/kernel_stats_default.csv
cp $O/kt_single/*/*kernel_stats.csv $O/kernel_stats_single.csv
# keep the merge-back small
rm -rf $O/kt_default $O/kt_single $O/pmc_fetch/*/*kernel_trace.csv $O/pmc_write/*/*kernel_trace.csv
ls -la $O
