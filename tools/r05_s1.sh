#!/bin/bash
# round 5, session 1: the sorted commit (k_commit_records + k_splat_sorted) and the busiest-first optimiser against the previous scheme
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "tuning or golden or learned_fraction or round_hook or room_stand_in or stepwise or mask or kitchen_improved or glossy" 2>&1 | tail -5
export PPG_AB_KERNELS=1
bash tools/ab.sh r05_s1_20 1 20 "-|" "-|PPG_NO_SORTED_COMMIT=1 PPG_ADAM_UNORDERED=1" "-|PPG_ADAM_UNORDERED=1"
bash tools/ab.sh r05_s1_127 1 127 "-|" "-|PPG_NO_SORTED_COMMIT=1 PPG_ADAM_UNORDERED=1" "-|PPG_ADAM_UNORDERED=1"
