#!/bin/bash
# round-2 GPU session 1: parity suite, then KITCHEN-geometry timings with the new round / tail structure
set -x
mkdir -p gpurun_out/s1
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/s1/pytest.log
tail -5 gpurun_out/s1/pytest.log
K="--scene-file scratch/kitchen-improved.ppgs --constant-env 12,12,12 --size-override --width 1280 --height 720 --no-cpu"
timeout 300 python bench.py $K --steps 127 --warmup 5 > gpurun_out/s1/kitchen127.json 2> gpurun_out/s1/kitchen127.err; tail -c 3000 gpurun_out/s1/kitchen127.json
timeout 300 python bench.py $K --steps 20 --warmup 5 > gpurun_out/s1/kitchen20.json 2> gpurun_out/s1/kitchen20.err; tail -c 1500 gpurun_out/s1/kitchen20.json
for T in 32768 524288; do
PPG_TAIL_THRESHOLD=$T timeout 300 python bench.py $K --steps 127 --warmup 5 --no-roofline > gpurun_out/s1/kitchen127_t$T.json 2>&1; tail -c 600 gpurun_out/s1/kitchen127_t$T.json
done
timeout 300 python bench.py --steps 255 --warmup 3 --no-cpu > gpurun_out/s1/cbox255.json 2> gpurun_out/s1/cbox255.err; tail -c 1500 gpurun_out/s1/cbox255.json
timeout 300 python bench.py --scene room --spp 1 --steps 127 --no-cpu > gpurun_out/s1/room127.json 2> gpurun_out/s1/room.err; tail -c 1500 gpurun_out/s1/room127.json
