#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export PPG_AB_TESTS="kitchen_improved_against_oracle or tuning_switches or torus_class or room_stand_in or cancel_from_another"
bash $R/tools/ab.sh r04_s18a 2 20 "-|PPG_PATH_LAYOUT=soa" "-|"
