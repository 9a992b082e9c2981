#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export PPG_AB_TESTS="kitchen_improved_against_oracle or spaceship_improved_against_oracle or glossy_plastic or rough_plastic or mask_bsdf or null_component or textures or analytic_spheres"
bash $R/tools/ab.sh r04_s5a 3 20 "-|PPG_NO_SPLIT=1" "-|PPG_NO_SORT_FIRST=1" "-|"
unset PPG_AB_TESTS
bash $R/tools/ab.sh r04_s5b 2 127 "-|PPG_NO_SPLIT=1" "-|PPG_NO_SORT_FIRST=1" "-|"
