#!/bin/bash
set -x
mkdir -p gpurun_out/s13
export TMPDIR=/tmp
cd /tmp
timeout 1500 python $GRAFT_REPO_ROOT/tools/collect_profiles_r02.py stats traffic wait > $GRAFT_REPO_ROOT/gpurun_out/s13/collect.log 2>&1
tail -30 $GRAFT_REPO_ROOT/gpurun_out/s13/collect.log
cd $GRAFT_REPO_ROOT
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
timeout 300 python bench.py --force-dist --steps 20 --warmup 5 $Q > gpurun_out/s13/bench_force_dist.json 2> gpurun_out/s13/bench_force_dist.log
timeout 300 python bench.py --scene room --steps 127 --warmup 5 $Q > gpurun_out/s13/room.json 2>/dev/null
timeout 300 python bench.py --scene room --glossy --steps 127 --warmup 5 $Q > gpurun_out/s13/room_glossy.json 2>/dev/null
timeout 300 python bench.py --scene torus --steps 255 --warmup 5 $Q > gpurun_out/s13/torus.json 2>/dev/null
timeout 300 python bench.py --scene-file scratch/spaceship.ppgs --steps 1023 --warmup 5 $Q > gpurun_out/s13/ship_640.json 2>/dev/null
