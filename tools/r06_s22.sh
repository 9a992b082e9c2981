#!/bin/bash
# GPU box: the BSDF-type sort of the queue slices inside k_trace — parity, then A/B against k_sort_slices as a launch of its own
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s22
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or room or kitchen or stragglers or spaceship or material or tuning or full_size or coincident" 2>&1 | tail -4
PPG_AB_KERNELS=1 tools/ab.sh r06_s22/ab20 2 20 "-|PPG_SORT_KERNEL=1" "-|"
PPG_AB_KERNELS=1 tools/ab.sh r06_s22/ab127 1 127 "-|PPG_SORT_KERNEL=1" "-|"
tools/ab.sh r06_s22/ab1023 1 1023 "-|PPG_SORT_KERNEL=1" "-|"
