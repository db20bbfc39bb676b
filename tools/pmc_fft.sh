cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_fft; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace -d $O/a --output-format csv -- python3 $R/tools/run_entry.py fft22 4 > $O/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY --kernel-trace -d $O/b --output-format csv -- python3 $R/tools/run_entry.py fft22 4 > $O/b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d $O/c --output-format csv -- python3 $R/tools/run_entry.py fft22 4 > $O/c.log 2>&1
cd $R && python tools/pmc_summary.py $O/a $O/b $O/c > gpurun_out/r2_pmc_fft.csv; rm -rf $O/*/*/*kernel_trace.csv
