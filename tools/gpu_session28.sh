#!/bin/bash
set -x
mkdir -p gpurun_out/s28
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
L=$PWD/practical-path-guiding_amd/lib
timeout 300 python bench.py --scene cbox --steps 255 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s28/cbox_base.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_w5.so timeout 300 python bench.py --scene cbox --steps 255 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s28/cbox_w5.json 2>/dev/null
timeout 300 python bench.py --scene torus --steps 1023 --warmup 5 $Q > gpurun_out/s28/torus_1023.json 2>/dev/null
timeout 300 python tools/scene_run.py scratch/spaceship.ppgs --sizes 1920x1080 --spp 1023 --parity "" --cpu-spp 0 -P sampleCombination=inversevar -P bsdfSamplingFractionLoss=kl -P spatialFilter=stochastic -P directionalFilter=box -P sTreeThreshold=4000 -P sppPerPass=1 --out gpurun_out/s28 > gpurun_out/s28/ship_improved.log 2>&1
tail -5 gpurun_out/s28/ship_improved.log
timeout 300 python tools/kitchen_error_probe.py 2400 > gpurun_out/s28/kitchen_2400.log 2>&1
tail -2 gpurun_out/s28/kitchen_2400.log
