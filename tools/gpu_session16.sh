#!/bin/bash
set -x
mkdir -p gpurun_out/s16
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
for C in "262144 16" "393216 16" "524288 16" "262144 8" "524288 8" "786432 12"; do
set -- $C
PPG_TAIL_MIN=$1 PPG_TAIL_DIV=$2 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s16/k127_min$1_div$2.json 2>/dev/null
PPG_TAIL_MIN=$1 PPG_TAIL_DIV=$2 timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s16/k20_min$1_div$2.json 2>/dev/null
done
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s16/k127_auto.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s16/k20_auto.json 2>/dev/null
