#!/bin/bash
# round 5, session 3: the optimiser's end of round beside the next batch's start (PPG_NO_ASIDE = without); counters of the new commit kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "tuning or golden or learned_fraction or round_hook or room_stand_in or stepwise or mask or kitchen_improved or cancel or cpp_ or rccl or sharded_contexts or time_budget" 2>&1 | tail -5
bash tools/ab.sh r05_s3_20 2 20 "-|" "-|PPG_NO_ASIDE=1"
bash tools/ab.sh r05_s3_127 1 127 "-|" "-|PPG_NO_ASIDE=1"
bash tools/ab.sh r05_s3_1023 1 1023 "-|" "-|PPG_NO_ASIDE=1"
cd /tmp; export TMPDIR=/tmp
timeout 900 python $R/tools/collect_profiles_r05.py traffic wait 2>&1 | tail -60
