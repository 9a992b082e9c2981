#!/bin/bash
# GPU box: k_shade<common> at four waves a SIMD (128 VGPRs, 75 spilled)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s18
PPG_AB_KERNELS=1 tools/ab.sh r06_s18/ab20 2 20 "-|" "libppg_hip_sc4.so|"
PPG_AB_KERNELS=1 tools/ab.sh r06_s18/ab127 1 127 "-|" "libppg_hip_sc4.so|"
