#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/ab.sh r04_s12a 3 20 "libppg_hip_v5.so|" "-|"
