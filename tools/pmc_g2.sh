# SQ counters of the G2 variable-base MSM kernels (and the G1 ones beside them, for the ratio)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_g2; rm -rf $O; mkdir -p $O
for what in var_g2 var_g1; do
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --kernel-trace -d $O/${what}_1 --output-format csv -- python3 $R/tools/run_entry.py $what 3 > $O/${what}_1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $O/${what}_2 --output-format csv -- python3 $R/tools/run_entry.py $what 3 > $O/${what}_2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace -d $O/${what}_3 --output-format csv -- python3 $R/tools/run_entry.py $what 3 > $O/${what}_3.log 2>&1
done
cd $R && python tools/pmc_summary.py $O/var_g2_1 $O/var_g2_2 $O/var_g2_3 > gpurun_out/r2_pmc_g2.csv && python tools/pmc_summary.py $O/var_g1_1 $O/var_g1_2 $O/var_g1_3 > gpurun_out/r2_pmc_g1.csv; rm -rf $O
