#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/sched_probe.py --sched p3 --depth 4 --ramp 2>&1 | grep -v amdgpu.ids
cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -20
cat /sys/class/drm/card*/device/power_dpm_force_performance_level 2>&1 | head
