#!/bin/bash
# GPU box: k_tail's compile-time thresholds, combinations
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s17
V='-| libppg_hip_x1.so| libppg_hip_x2.so| libppg_hip_x3.so| libppg_hip_x4.so|'
tools/ab.sh r06_s17/ab20 3 20 $V
tools/ab.sh r06_s17/ab127 1 127 $V
