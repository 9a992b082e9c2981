#!/bin/bash
set -x
mkdir -p gpurun_out/s29
export TMPDIR=/tmp
timeout 600 python tools/scene_run.py scratch/kitchen-improved.ppgs --parity 160x90 --parity-spp 31 --sizes 1280x720,700x400 --spp 127 --cpu-spp 0 --out gpurun_out/s29 > gpurun_out/s29/kitchen.log 2>&1
tail -4 gpurun_out/s29/kitchen.log
timeout 600 python tools/scene_run.py scratch/spaceship.ppgs --parity 320x180 --parity-spp 15 --sizes 640x360,1920x1080 --spp 1023 --cpu-spp 0 --out gpurun_out/s29 > gpurun_out/s29/spaceship.log 2>&1
tail -4 gpurun_out/s29/spaceship.log
rm -f gpurun_out/s29/*.npy
