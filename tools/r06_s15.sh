#!/bin/bash
# GPU box: k_tail's traversals suspended when few lanes are left (trace_closest4_resume) — parity, then A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s15
timeout 900 python -m pytest tests -m gpu -x -q -k "room or kitchen or bvh or stragglers or real or spaceship or full_size or two_ranks or unbounded" 2>&1 | tail -5 > gpurun_out/r06_s15/tests.log
cat gpurun_out/r06_s15/tests.log
PPG_AB_KERNELS=1 tools/ab.sh r06_s15/ab20 2 20 "libppg_hip_su0.so|" "-|" "libppg_hip_su4.so|" "libppg_hip_su16.so|"
tools/ab.sh r06_s15/ab127 1 127 "libppg_hip_su0.so|" "-|" "libppg_hip_su4.so|" "libppg_hip_su16.so|"
