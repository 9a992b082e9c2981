#!/bin/bash
set -x
mkdir -p gpurun_out/s30
export TMPDIR=/tmp
for F in 64 128 256; do
PPG_FINAL_BATCH=$F timeout 300 python tools/kitchen_error_probe.py 2400 > gpurun_out/s30/probe_fb$F.log 2>&1
grep -o '"seconds": \[[^]]*\]' gpurun_out/s30/probe_fb$F.log
done
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
PPG_FINAL_BATCH=256 timeout 300 python bench.py --steps 511 --warmup 5 $Q > gpurun_out/s30/k511_fb256.json 2>/dev/null
timeout 300 python bench.py --steps 511 --warmup 5 $Q > gpurun_out/s30/k511_fb64.json 2>/dev/null
