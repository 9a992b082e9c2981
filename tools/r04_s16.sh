#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export PPG_AB_TESTS="kitchen_improved_against_oracle"
bash $R/tools/ab.sh r04_s16a 2 20 "-|" "-|PPG_PATH_LAYOUT=pack" "-|PPG_PATH_LAYOUT=aos"
unset PPG_AB_TESTS
bash $R/tools/ab.sh r04_s16b 2 127 "-|" "-|PPG_PATH_LAYOUT=pack" "-|PPG_PATH_LAYOUT=aos"
