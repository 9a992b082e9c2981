#!/bin/bash
set -x
mkdir -p gpurun_out/s12
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s12/pytest.log 2>&1
tail -5 gpurun_out/s12/pytest.log
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
L=$PWD/practical-path-guiding_amd/lib
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s12/k127_a1.json 2>/dev/null
for V in ls22 ls20 w4; do
PPG_HIP_LIB=$L/libppg_hip_$V.so timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s12/k127_$V.json 2>/dev/null
done
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s12/k127_a2.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s12/k20_a.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_ls22.so timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s12/k20_ls22.json 2>/dev/null
timeout 300 python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s12/k127_a_kernels.json 2>/dev/null
timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s12/ship_a.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_ls22.so timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 $Q > gpurun_out/s12/ship_ls22.json 2>/dev/null
timeout 300 python bench.py --scene cbox --steps 255 --warmup 5 $Q > gpurun_out/s12/cbox_a.json 2>/dev/null
