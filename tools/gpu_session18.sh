#!/bin/bash
set -x
mkdir -p gpurun_out/s18
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
L=$PWD/practical-path-guiding_amd/lib
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s18/k127_base1.json 2>/dev/null
for V in lv8 lv24 lv32 bs4 bs12 ls20; do
PPG_HIP_LIB=$L/libppg_hip_$V.so timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s18/k127_$V.json 2>/dev/null
done
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s18/k127_base2.json 2>/dev/null
for B in 2048 8192; do
PPG_BLOCKS=$B timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s18/k127_blocks$B.json 2>/dev/null
done
PPG_NO_OVERLAP=1 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s18/k127_nooverlap.json 2>/dev/null
PPG_NO_SORT=1 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s18/k127_nosort.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_ls20.so timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 $Q > gpurun_out/s18/ship_ls20.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_lv8.so timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 $Q > gpurun_out/s18/ship_lv8.json 2>/dev/null
timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 $Q > gpurun_out/s18/ship_base.json 2>/dev/null
