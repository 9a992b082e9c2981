#!/bin/bash
set -x
mkdir -p gpurun_out/s17
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s17/pytest.log 2>&1
tail -4 gpurun_out/s17/pytest.log
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
L=$PWD/practical-path-guiding_amd/lib
for i in 1 2; do
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s17/k127_lds_$i.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_nobox.so timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s17/k127_nobox_$i.json 2>/dev/null
done
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s17/k20_lds.json 2>/dev/null
PPG_HIP_LIB=$L/libppg_hip_nobox.so timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s17/k20_nobox.json 2>/dev/null
for C in "1048576 8" "1572864 6" "1048576 12"; do
set -- $C
PPG_TAIL_MIN=$1 PPG_TAIL_DIV=$2 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s17/k127_min$1_div$2.json 2>/dev/null
PPG_TAIL_MIN=$1 PPG_TAIL_DIV=$2 timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s17/k20_min$1_div$2.json 2>/dev/null
done
timeout 300 python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s17/k127_kernels.json 2>/dev/null
