#!/bin/bash
cd $GRAFT_REPO_ROOT
for h in 3 5 7 11 3; do echo "== OZK_COPY_HELPERS=$h"; OZK_COPY_HELPERS=$h python tools/host_path.py 20 2>&1 | grep "var_msm_host G1\|fixed_batch_msm_host G1\|fft_host\|compact_host G2" | cut -c1-46,62-100; done
