#!/bin/bash
set -x
mkdir -p gpurun_out/s15
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
timeout 300 python -X faulthandler bench.py --force-dist --steps 20 --warmup 5 $Q > gpurun_out/s15/bench_force_dist.json 2> gpurun_out/s15/bench_force_dist.log; echo "rc=$?" >> gpurun_out/s15/bench_force_dist.log
tail -8 gpurun_out/s15/bench_force_dist.log
for T in 16384 32768 65536 131072 262144; do
PPG_TAIL_THRESHOLD=$T timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s15/k127_thr$T.json 2>/dev/null
PPG_TAIL_THRESHOLD=$T timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s15/k20_thr$T.json 2>/dev/null
done
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s15/k127_auto.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s15/k20_auto.json 2>/dev/null
