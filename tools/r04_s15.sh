#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/ab.sh r04_s15a 2 127 "-|" "libppg_hip_c4.so|"
