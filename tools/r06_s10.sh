#!/bin/bash
# GPU box: pair compaction in k_trace AND k_tail's crowd phase, vote on the number of pairs — parity, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s10
timeout 900 python -m pytest tests -m gpu -x -q -k "room or kitchen or bvh or stragglers or real or spaceship or full_size or two_ranks" 2>&1 | tail -5 > gpurun_out/r06_s10/tests.log
cat gpurun_out/r06_s10/tests.log
PPG_AB_KERNELS=1 tools/ab.sh r06_s10/ab20 2 20 "libppg_hip_np.so|" "-|" "libppg_hip_tp0.so|" "libppg_hip_pv32.so|" "libppg_hip_pv64.so|" "libppg_hip_tv16.so|" "libppg_hip_tv48.so|" "libppg_hip_s12.so|"
PPG_AB_KERNELS=1 tools/ab.sh r06_s10/ab127 1 127 "libppg_hip_np.so|" "-|" "libppg_hip_tp0.so|" "libppg_hip_pv32.so|" "libppg_hip_pv64.so|" "libppg_hip_tv16.so|" "libppg_hip_tv48.so|" "libppg_hip_s12.so|"
