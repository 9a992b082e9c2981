#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export PPG_AB_TESTS="kitchen_improved_against_oracle or spaceship_improved_against_oracle"
bash $R/tools/ab.sh r04_s10a 3 20 "libppg_hip_v5.so|" "-|"
unset PPG_AB_TESTS
bash $R/tools/ab.sh r04_s10b 2 127 "libppg_hip_v5.so|" "-|"
