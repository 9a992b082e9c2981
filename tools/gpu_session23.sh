#!/bin/bash
set -x
mkdir -p gpurun_out/s23
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s23/pytest.log 2>&1
tail -4 gpurun_out/s23/pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s23/k20_$i.json 2>/dev/null
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s23/k127_$i.json 2>/dev/null
done
timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 --warmup 5 $Q > gpurun_out/s23/ship.json 2>/dev/null
timeout 300 python bench.py --scene room --steps 127 --warmup 5 $Q > gpurun_out/s23/room.json 2>/dev/null
