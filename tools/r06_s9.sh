#!/bin/bash
# GPU box: (ray, triangle)-pair compaction in k_trace — parity first, then A/B against the lane-per-leaf build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s9
timeout 900 python -m pytest tests -m gpu -x -q -k "room or kitchen or bvh or stragglers or real or spaceship or full_size" 2>&1 | tail -5 > gpurun_out/r06_s9/tests.log
cat gpurun_out/r06_s9/tests.log
PPG_AB_KERNELS=1 tools/ab.sh r06_s9/ab20 2 20 "libppg_hip_nopairs.so|" "-|" "libppg_hip_s16.so|" "libppg_hip_v8.so|" "libppg_hip_v32.so|"
PPG_AB_KERNELS=1 tools/ab.sh r06_s9/ab127 1 127 "libppg_hip_nopairs.so|" "-|" "libppg_hip_s16.so|" "libppg_hip_v8.so|" "libppg_hip_v32.so|"
