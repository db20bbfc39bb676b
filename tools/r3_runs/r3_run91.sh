#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $R/gpurun_out/pp --output-format csv -- python3 $R/tools/groth16_prove.py 20 3 > $R/gpurun_out/r3_prove_prof.log 2>&1
cd $R && python tools/prof_timeline.py gpurun_out/pp k_r1cs_eval 20 > gpurun_out/r3_timeline_prove.txt; rm -rf gpurun_out/pp
wc -l gpurun_out/r3_timeline_prove.txt; tail -60 gpurun_out/r3_timeline_prove.txt | cut -c1-140
