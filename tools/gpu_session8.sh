#!/bin/bash
set -x
mkdir -p gpurun_out/s8
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s8/pytest.log 2>&1
tail -5 gpurun_out/s8/pytest.log
# C++ RCCL reducer, real communicator on one rank
R=practical-path-guiding_amd/bin/ppg_render
rm -f /tmp/ncclid
NCCL_DEBUG=INFO timeout 300 $R --rank 0 --world 1 --nccl-id /tmp/ncclid -o gpurun_out/s8/kitchen_rccl.pfm -D budget=31 scratch/kitchen-improved.ppgs > gpurun_out/s8/rccl_cpp.log 2>&1
tail -4 gpurun_out/s8/rccl_cpp.log
timeout 300 $R -q -o gpurun_out/s8/kitchen_plain.pfm -D budget=31 scratch/kitchen-improved.ppgs > gpurun_out/s8/plain_cpp.log 2>&1
cmp gpurun_out/s8/kitchen_rccl.pfm gpurun_out/s8/kitchen_plain.pfm && echo "RCCL world=1 picture identical to un-sharded picture" >> gpurun_out/s8/rccl_cpp.log
rm -f gpurun_out/s8/*.pfm
NCCL_DEBUG=INFO timeout 300 python bench.py --force-dist --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > gpurun_out/s8/bench_force_dist.json 2> gpurun_out/s8/bench_force_dist.log
timeout 600 python tools/kitchen_error_probe.py > gpurun_out/s8/probe.log 2>&1
tail -3 gpurun_out/s8/probe.log
