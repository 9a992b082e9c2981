#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export PPG_AB_TESTS="kitchen_improved_against_oracle or large_scene_bvh or torus_class or room_stand_in"
bash $R/tools/ab.sh r04_s6a 3 20 "libppg_hip_v3.so|" "libppg_hip_v4.so|" "-|PPG_NO_TOPCUT=1" "-|"
