#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pf_fft --output-format csv -- python3 $R/tools/run_entry.py fft22 10 > /dev/null 2>&1
python3 $R/tools/stats_grep.py $R/gpurun_out/pf_fft fft_pass
rm -rf $R/gpurun_out/pf_fft
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $R/gpurun_out/pmc_fft --output-format csv -- python3 $R/tools/run_entry.py fft22 3 > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_fft 2>&1 | grep "fft_pass\|kernel," | cut -c1-300
rm -rf $R/gpurun_out/pmc_fft
