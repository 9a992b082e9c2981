#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export PPG_AB_TESTS="kitchen_improved_against_oracle or room_stand_in or torus_class"
bash $R/tools/ab.sh r04_s8a 3 20 "libppg_hip_v5.so|" "-|PPG_NO_TAIL_COMMIT=1" "-|"
