#!/bin/bash
# GPU box: k_trace pair compaction, the vote threshold at eight waves; 1023 passes against the round-5 kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s12
V='libppg_hip_a.so| -| libppg_hip_pv16.so| libppg_hip_pv24.so| libppg_hip_pv48.so| libppg_hip_s10.so|'
tools/ab.sh r06_s12/ab20 2 20 $V
tools/ab.sh r06_s12/ab127 2 127 $V
tools/ab.sh r06_s12/ab1023 1 1023 "libppg_hip_a.so|" "-|"
