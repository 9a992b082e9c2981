#!/bin/bash
# GPU box, round 6, session 8: k_splat_sorted with short runs straight into the pool — parity, then A/B against the previous build
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_s8
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "golden or stragglers or tuning_switches or improved or two_ranks or kitchen or image_region or sorted_splat or optim" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
PPG_AB_KERNELS=1 tools/ab.sh r06_ab11 2 20 "-|" "libppg_hip_prev.so|" 2>&1 | tail -3 | cut -c1-700
tools/ab.sh r06_ab12 1 127 "-|" "libppg_hip_prev.so|" 2>&1 | tail -3
tools/ab.sh r06_ab13 1 1023 "-|" "libppg_hip_prev.so|" 2>&1 | tail -3
cd /tmp
python $R/tools/region_cost_probe.py 2>&1 | grep "passes 20"
