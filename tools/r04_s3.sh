#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
export PPG_AB_TESTS="kitchen_improved_against_oracle or spaceship_improved_against_oracle or room_stand_in or torus_class or large_scene_bvh or analytic_spheres or null_component or next_event_estimation_bvh"
bash $R/tools/ab_lib.sh r04_s3a 3 20 libppg_hip_r03.so libppg_hip_v1.so -
unset PPG_AB_TESTS
bash $R/tools/ab_lib.sh r04_s3b 2 127 libppg_hip_r03.so libppg_hip_v1.so -
