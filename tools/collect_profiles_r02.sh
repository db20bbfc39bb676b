#!/bin/bash
# Round-2 evidence (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats of the same
# commands and of every other entry point, PMC HBM + SQ counters of the dominant kernel (separate passes).
# Outputs under gpurun_out/evidence2/; copy what should be judged to profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence2
rm -rf $O && mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --in-flight 1 --no-cpu-baseline > $O/bench_in_flight_1.json 2> $O/bench_in_flight_1.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt_default --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only > $O/kt_default.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_single --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 10 --in-flight 1 > $O/kt_single.log 2>&1
echo "bench traces done"
for w in fft22 fixed_g1 fixed_g2 var_g2 qap21; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$w --output-format csv -- python3 $R/tools/run_entry.py $w 10 > $O/kt_$w.log 2>&1
  cp $O/kt_$w/*/*kernel_stats.csv $O/kernel_stats_$w.csv
  echo "$w done"
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 4 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 4 --warmup 1 > $O/pmc_write.log 2>&1
echo "hbm pmc done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace -d $O/pmc_sq1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 4 --warmup 1 > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq2 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 4 --warmup 1 > $O/pmc_sq2.log 2>&1
echo "sq pmc done"
cd $R
python tools/prof_summary.py $O/kt_single > $O/timeline_single.txt
python tools/prof_pipeline.py $O/kt_default > $O/timeline_pipelined.txt
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_hbm_summary.csv
python tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 > $O/pmc_sq_summary.csv
cp $O/kt_default/*/*kernel_stats.csv $O/kernel_stats_default.csv
cp $O/kt_single/*/*kernel_stats.csv $O/kernel_stats_single.csv
rm -rf $O/kt_* $O/pmc_*/*/*kernel_trace.csv
du -sh $O; ls $O
