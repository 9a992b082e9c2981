#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_s2; mkdir -p $OUT
export PPG_AB_TESTS="kitchen_improved_against_oracle or room_stand_in or torus_class or large_scene_bvh or analytic_spheres"
bash $R/tools/ab_lib.sh r04_s2 3 20 libppg_hip_r03.so -
unset PPG_AB_TESTS
cd /tmp
PPG_HIP_LIB=$R/practical-path-guiding_amd/lib/libppg_hip_probe.so python $R/bench.py --steps 20 --warmup 0 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/probe20.json 2> $OUT/probe20.err
grep -h "ppg probe" $OUT/probe20.err | tail -1
