#!/bin/bash
set -x
mkdir -p gpurun_out/s5
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s5/pytest.log 2>&1; tail -3 gpurun_out/s5/pytest.log
B="python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary"
timeout 300 $B > gpurun_out/s5/k127.json 2>/dev/null; tail -c 1500 gpurun_out/s5/k127.json
timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s5/ship255.json 2>/dev/null; tail -c 1500 gpurun_out/s5/ship255.json
timeout 300 python bench.py --scene room --steps 127 --no-cpu --no-rmse --no-secondary > gpurun_out/s5/room127.json 2>/dev/null; tail -c 1200 gpurun_out/s5/room127.json
timeout 300 python bench.py --scene cbox --steps 255 --no-cpu --no-rmse --no-secondary > gpurun_out/s5/cbox255.json 2>/dev/null; tail -c 1200 gpurun_out/s5/cbox255.json
