#!/bin/bash
# GPU box: k_trace pair compaction at equal occupancy — stack rows / forced 8 waves; totals of untimed renders
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s11
V='libppg_hip_a.so| libppg_hip_b.so| libppg_hip_c.so| -| libppg_hip_e.so| libppg_hip_f.so| libppg_hip_g.so|'
PPG_AB_TESTS="room or kitchen or bvh or stragglers" tools/ab.sh r06_s11/t 0 20 "libppg_hip_e.so|" "libppg_hip_f.so|" 2>&1 | grep -v "^v"
tools/ab.sh r06_s11/ab20 3 20 $V
tools/ab.sh r06_s11/ab127 2 127 $V
