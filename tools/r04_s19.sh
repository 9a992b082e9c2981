#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/ab.sh r04_s19a 2 127 "-|" "-|PPG_TAIL_DIV=24" "-|PPG_TAIL_DIV=48" "-|PPG_TAIL_DIV=6" "-|PPG_TAIL_DIV=24 PPG_TAIL_MIN=1048576"
