#!/bin/bash
# same-box A/B of two builds of the library (OZK_LIB_PATH): bench.py in the driver's form and over 100 steps, alternating
#   gpurun -- "bash tools/ab_builds.sh octopuszk_amd/_ab/libozk_hip_OLD.so octopuszk_amd/libozk_hip.so [reps]"
A=$1; B=$2; R=${3:-3}
for i in $(seq 1 $R); do
  for lib in $A $B; do
    for args in "--steps 20 --warmup 5" "--steps 100 --warmup 5"; do
      OZK_LIB_PATH=$PWD/$lib python bench.py --gpus 1 $args --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=j['roofline']
print('%-44s steps %3d  %7.1f Mscalar-mul/s  %.4f ms/step  L1 in schedule %.3f alone %.3f  clk %.0f / %.0f' % ('$lib', j['steps'], j['value'], j['ms_per_step'], r['kernel_ms']['median'], r['kernel_ms_alone']['median'], r['shader_clock_mhz']['in_schedule']['median'], r['shader_clock_mhz']['alone']['median']))"
    done
  done
done
