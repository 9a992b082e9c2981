#!/bin/bash
set -x
mkdir -p gpurun_out/s19
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
L=$PWD/practical-path-guiding_amd/lib
PPG_HIP_LIB=$L/libppg_hip_c1.so PPG_BLOCKS=8192 timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s19/pytest_c1.log 2>&1
tail -4 gpurun_out/s19/pytest_c1.log
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s19/k127_base.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_c1.so timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s19/k127_c1.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_c1.so PPG_BLOCKS=8192 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s19/k127_c1_b8192.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_c2.so PPG_BLOCKS=8192 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s19/k127_c2_b8192.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_c1.so PPG_BLOCKS=8192 timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s19/k20_c1_b8192.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_c1.so PPG_BLOCKS=16384 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s19/k127_c1_b16384.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s19/k20_base.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_c1.so PPG_BLOCKS=8192 timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 $Q > gpurun_out/s19/ship_c1_b8192.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_c1.so PPG_BLOCKS=8192 timeout 300 python bench.py --scene cbox --steps 255 --warmup 5 $Q > gpurun_out/s19/cbox_c1_b8192.json 2>/dev/null
timeout 300 python bench.py --scene cbox --steps 255 --warmup 5 $Q > gpurun_out/s19/cbox_base.json 2>/dev/null
