#!/bin/bash
set -x
mkdir -p gpurun_out/s24
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
L=$PWD/practical-path-guiding_amd/lib
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s24/k20_pre_$i.json 2>/dev/null
PPG_NO_PREALLOC=1 timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s24/k20_nopre_$i.json 2>/dev/null
done
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s24/k127_pre.json 2>/dev/null
PPG_NO_PREALLOC=1 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s24/k127_nopre.json 2>/dev/null
for V in tv0 tv4 tv8 tv16; do
PPG_HIP_LIB=$L/libppg_hip_$V.so timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s24/k20_$V.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_$V.so timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s24/k127_$V.json 2>/dev/null
done
