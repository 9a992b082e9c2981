#!/bin/bash
# GPU box: k_tail's compile-time thresholds with the suspended traversals
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s16
V='-| libppg_hip_cm3.so| libppg_hip_cm10.so| libppg_hip_lv4.so| libppg_hip_lv16.so| libppg_hip_su8m.so|'
tools/ab.sh r06_s16/ab20 2 20 $V
tools/ab.sh r06_s16/ab127 1 127 $V
