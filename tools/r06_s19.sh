#!/bin/bash
# GPU box: k_tail<FULL> at 4 waves a SIMD (128 VGPRs, 304 spilled) and at 2 (201, none) against 3 (168, 94 spilled)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s19
tools/ab.sh r06_s19/ab20 3 20 "-|" "libppg_hip_tw4.so|" "libppg_hip_tw2.so|"
tools/ab.sh r06_s19/ab127 1 127 "-|" "libppg_hip_tw4.so|" "libppg_hip_tw2.so|"
