#!/bin/bash
# Where a small MSM's time goes (n = 2^10: the verifier's / the Java prover's primary-input size): per-kernel timeline
# of one lone MSM, then the effect of the plan knobs.   gpurun -- "bash tools/small_n_probe.sh"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/sn --output-format csv -- python3 $R/tools/msm_once.py 10 > /dev/null 2>&1
cd $R
python tools/prof_summary.py gpurun_out/sn
rm -rf gpurun_out/sn
for kn in "" "OZK_MSM_L1=8" "OZK_MSM_L1=16" "OZK_MSM_L1=8 OZK_MSM_C=7" "OZK_MSM_L1=8 OZK_MSM_C=6" "OZK_MSM_L1=8 OZK_MSM_C=10" "OZK_MSM_L1=8 OZK_MSM_S_LAT=4" "OZK_MSM_L1=8 OZK_MSM_S_LAT=16" "OZK_MSM_L1=8 OZK_MSM_FIN_MAX=8"; do
  echo "== $kn"; env $kn python tools/size_sweep.py 10 11 2>&1 | grep "n=" | head -2
done
