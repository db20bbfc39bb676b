#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for v in "" dbgtw "" dbgtw; do
  if [ -z "$v" ]; then unset OZK_LIB_PATH; else export OZK_LIB_PATH=$R/octopuszk_amd/libozk_$v.so; fi
  echo -n "lib[$v] "; python tools/run_entry.py fft22 30 2>&1 | grep -v amdgpu.ids | tail -1
  echo -n "lib[$v] "; python tools/run_entry.py qap21 20 2>&1 | grep -v amdgpu.ids | tail -1
done
export OZK_LIB_PATH=$R/octopuszk_amd/libozk_dbgtw.so
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pf_dbg --output-format csv -- python3 $R/tools/run_entry.py fft22 10 > /dev/null 2>&1
python3 $R/tools/stats_grep.py $R/gpurun_out/pf_dbg fft_pass; rm -rf $R/gpurun_out/pf_dbg
