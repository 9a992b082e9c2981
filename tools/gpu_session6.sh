#!/bin/bash
set -x
mkdir -p gpurun_out/s6
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_textures.py -m gpu -x -q -k "large_scene or bvh or spheres or environment or null or torus or textures or stepwise or full_size" > gpurun_out/s6/pytest.log 2>&1; tail -3 gpurun_out/s6/pytest.log
B="python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary"
timeout 300 $B > gpurun_out/s6/k127.json 2>/dev/null; tail -c 1300 gpurun_out/s6/k127.json
timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s6/ship255.json 2>/dev/null; tail -c 1300 gpurun_out/s6/ship255.json
timeout 300 python bench.py --scene room --steps 127 --no-cpu --no-rmse --no-secondary > gpurun_out/s6/room127.json 2>/dev/null; tail -c 1000 gpurun_out/s6/room127.json
