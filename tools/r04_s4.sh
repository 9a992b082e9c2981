#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
bash $R/tools/ab_lib.sh r04_s4a 3 20 - libppg_hip_r03.so libppg_hip_v1.so
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temp" | head
