#!/bin/bash
# round 5, session 4: k_commit_records with four lanes per path against one (libppg_hip_lane1.so)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "tuning or golden or learned_fraction or room_stand_in or stepwise or kitchen_improved or sharded_contexts" 2>&1 | tail -5
export PPG_AB_KERNELS=1
bash tools/ab.sh r05_s4_127 1 127 "-|" "libppg_hip_lane1.so|"
bash tools/ab.sh r05_s4_1023 1 1023 "-|" "libppg_hip_lane1.so|"
unset PPG_AB_KERNELS
bash tools/ab.sh r05_s4_20 2 20 "-|" "libppg_hip_lane1.so|"
