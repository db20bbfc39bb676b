#!/bin/bash
# Round-4 evidence, ALL of it from one build in one gpurun call (run from the repo root):
#     gpurun --timeout 1200 -- "bash tools/collect_profiles_r04.sh $(git rev-parse --short HEAD)"
# then `python tools/publish_profiles_r04.py` copies gpurun_out/evidence4/r04_* into profiles/ (tracked) and refreshes
# profiles/traffic.json.  Every text / JSON file carries the commit; r04_MANIFEST.txt lists the files, the commit,
# the date and the device.  Contents: bench lines in the driver's form (--steps 20 --warmup 5) and at 100 steps, the
# two-stage schedule and the lone-MSM form; rocprofv3 --kernel-trace --stats of the bench command and of every other
# entry point; PMC HBM (FETCH_SIZE, WRITE_SIZE: separate passes) and SQ counters of the level-1 kernel; the
# steady-state timeline of the three-stage schedule; the end-to-end proof; the host-path table and call-time
# distributions; the kernel resource table.
set -o pipefail
COMMIT=${1:-unknown}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence4
rm -rf $O && mkdir -p $O
cd $R
hdr() { echo "# commit $COMMIT  $(date -u +%Y-%m-%dT%H:%MZ)  $1"; }
line() { python -c "
import json,sys
l=[x for x in open('$1') if x.startswith('{')]
j=json.loads(l[-1]); j['evidence_commit']='$COMMIT'
print(json.dumps(j))" > $2; }
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/b1.out 2> $O/b1.err && line $O/b1.out $O/r04_bench_driver_form.json
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/b2.out 2> $O/b2.err && line $O/b2.out $O/r04_bench_driver_form_rerun.json
python bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/b3.out 2> $O/b3.err && line $O/b3.out $O/r04_bench_100_steps.json
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --schedule pipeline > $O/b4.out 2> $O/b4.err && line $O/b4.out $O/r04_bench_two_stage.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 --schedule pipeline > $O/b5.out 2> $O/b5.err && line $O/b5.out $O/r04_bench_in_flight_1.json
echo "bench done $SECONDS s"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/kt_default --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 100 > $O/kt_default.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/kt_single --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 10 --in-flight 1 --schedule pipeline > $O/kt_single.log 2>&1
cp $O/kt_default/*/*kernel_stats.csv $O/r04_kernel_stats_default.csv
cp $O/kt_single/*/*kernel_stats.csv $O/r04_kernel_stats_single.csv
echo "bench traces done $SECONDS s"
for w in fft22 fixed_g1 fixed_g2 fixed_g1_rebuild fixed_g2_rebuild var_g2 qap21; do
  rocprofv3 --kernel-trace --stats -d $O/kt_$w --output-format csv -- python3 $R/tools/run_entry.py $w 10 > $O/kt_$w.log 2>&1
  cp $O/kt_$w/*/*kernel_stats.csv $O/r04_kernel_stats_$w.csv
  echo "$w done $SECONDS s"
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 4 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 4 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_fft --output-format csv -- python3 $R/tools/run_entry.py fft22 3 > $O/pmc_fetch_fft.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_fft --output-format csv -- python3 $R/tools/run_entry.py fft22 3 > $O/pmc_write_fft.log 2>&1
echo "hbm pmc done $SECONDS s"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace -d $O/pmc_sq1 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 4 --warmup 1 > $O/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_sq2 --output-format csv -- python3 $R/bench.py --no-cpu-baseline --timed-only --steps 4 --warmup 1 > $O/pmc_sq2.log 2>&1
echo "sq pmc done $SECONDS s"
cd $R
{ hdr "lone MSM 2^20 (bench --in-flight 1 --schedule pipeline), last step"; python tools/prof_summary.py $O/kt_single; } > $O/r04_timeline_single_msm.txt
{ hdr "three-stage schedule, steady state (bench --timed-only --steps 100 under rocprofv3 --kernel-trace)"; python tools/timeline_steady.py $O/kt_default 2 3; } > $O/r04_timeline_three_stage.txt
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/r04_pmc_hbm_summary.csv
python tools/pmc_summary.py $O/pmc_sq1 $O/pmc_sq2 > $O/r04_pmc_sq_summary.csv
python tools/pmc_summary.py $O/pmc_fetch_fft $O/pmc_write_fft > $O/r04_pmc_hbm_fft_summary.csv
{ hdr "serial Groth16 at 2^20 constraints (tools/groth16_prove.py 20 8)"; python tools/groth16_prove.py 20 8 2>&1 | grep -v amdgpu.ids; } > $O/r04_groth16_prove_2p20.txt
grep '^{' $O/r04_groth16_prove_2p20.txt | python -c "
import json,sys
j=json.loads(sys.stdin.read()); j['evidence_commit']='$COMMIT'; print(json.dumps(j))" > $O/r04_groth16_prove_2p20.json
{ hdr "JNI-shaped entry points from fresh pageable buffers (tools/host_path.py)"; python tools/host_path.py 2>&1 | grep -v amdgpu.ids; } > $O/r04_host_path.txt
{ hdr "fill and drain of the three-stage schedule: total = c0 + K * s over bursts of 10-200 MSMs (tools/sched_probe.py --fit)"; python tools/sched_probe.py --sched p3 --depth 4 --fit 2>&1 | grep -v amdgpu.ids | cut -c1-60; } > $O/r04_fill_drain_fit.txt
{ hdr "device-resident MSM time by size (tools/size_sweep.py)"; python tools/size_sweep.py 2>&1 | grep -v amdgpu.ids; } > $O/r04_size_sweep.txt
{ hdr "schedules (tools/sched_probe.py, 200 MSMs each)"; for a in "--sched p3 --depth 4" "--sched p2" "--sched p3 --depth 4 --prepared"; do python tools/sched_probe.py --reps 200 $a 2>&1 | grep -v amdgpu.ids; done; } > $O/r04_schedules.txt
{ hdr "Montgomery multiplication variants incl. J = FP64-FMA 52-bit limbs, EL = E in a level-1-sized loop body, E sustained with the shader clock (tools/ubench_mont.hip)"; (cd tools && ./ubench_mont) 2>&1 | grep -v amdgpu.ids; } > $O/r04_ubench_mont.txt
{ hdr "the level-1 formula alone, registers only (tools/ubench_madd.hip)"; tools/ubench_madd 2>&1 | grep -v amdgpu.ids; } > $O/r04_ubench_madd.txt
{ hdr "the level-1 loop rebuilt piece by piece on synthetic sorted arrays (tools/ubench_l1loop.hip)"; tools/ubench_l1loop 2>&1 | grep -v amdgpu.ids; } > $O/r04_ubench_l1loop.txt
{ hdr "level-1 kernel alone by digit distribution (tools/l1_probe.py)"; python tools/l1_probe.py 2>&1 | grep -v amdgpu.ids; } > $O/r04_l1_probe.txt
{ hdr "a 2^10-pair MSM: per-kernel timeline and plan knobs (tools/small_n_probe.sh)"; bash tools/small_n_probe.sh 2>&1 | grep -v amdgpu.ids; } > $O/r04_small_n_probe.txt
{ hdr "every call of the host entry points with the library's own account of its waits, page faults and context switches (tools/host_path.py --all-stats)"; python tools/host_path.py --all-stats --only=double --only=fixed_batch_msm_host 2>&1 | grep -v amdgpu.ids | cut -c1-260; } > $O/r04_host_path_allstats.txt
{ echo "commit $COMMIT"; date -u; python -c "import torch; print(torch.cuda.get_device_name(0))"; ls $O | grep r04_; } > $O/r04_MANIFEST.txt
rm -rf $O/kt_* $O/pmc_fetch $O/pmc_write $O/pmc_fetch_fft $O/pmc_write_fft $O/pmc_sq1 $O/pmc_sq2 $O/*.out
echo "all done $SECONDS s"; du -sh $O; ls $O
