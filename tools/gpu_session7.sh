#!/bin/bash
set -x
mkdir -p gpurun_out/s7
export TMPDIR=/tmp
B="python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary"
L=$PWD/practical-path-guiding_amd/lib
timeout 300 $B > gpurun_out/s7/v0.json 2>/dev/null
for V in v8 v16 v32; do
PPG_HIP_LIB=$L/libppg_hip_$V.so timeout 300 $B > gpurun_out/s7/$V.json 2>/dev/null
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s7/k20.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_v16.so timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s7/ship_v16.json 2>/dev/null
