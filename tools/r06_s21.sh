#!/bin/bash
# GPU box: shade_one with the S-tree's grid entry and leaf header fetched ahead of the material — parity, then A/B against the previous build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s21
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or cbox or room or kitchen or stragglers or spaceship or nee or material or sphere or env" 2>&1 | tail -4
PPG_AB_KERNELS=1 tools/ab.sh r06_s21/ab20 2 20 "libppg_hip_prev.so|" "-|"
PPG_AB_KERNELS=1 tools/ab.sh r06_s21/ab127 1 127 "libppg_hip_prev.so|" "-|"
