#!/bin/bash
# round 5, session 6: k_sort_slices with the per-triangle class table (PPG_NO_TRI_CLASS = without); the new reference-log test; smoke
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "tuning or reference_logs or kitchen_improved or golden or glossy or mask or textures or spheres" 2>&1 | tail -5
python __graft_entry__.py smoke 2>&1 | tail -3
export PPG_AB_KERNELS=1
bash tools/ab.sh r05_s6_127 1 127 "-|" "-|PPG_NO_TRI_CLASS=1"
bash tools/ab.sh r05_s6_1023 1 1023 "-|" "-|PPG_NO_TRI_CLASS=1"
unset PPG_AB_KERNELS
bash tools/ab.sh r05_s6_20 2 20 "-|" "-|PPG_NO_TRI_CLASS=1"
