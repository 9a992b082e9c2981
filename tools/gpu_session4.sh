#!/bin/bash
# round-2 GPU session 4: occupancy of the FULL shade / tail kernels (2, 3, 4 waves per SIMD) and the number of persistent workgroups
set -x
mkdir -p gpurun_out/s4
export TMPDIR=/tmp
B="python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary"
L=$PWD/practical-path-guiding_amd/lib
for W in w2 w3; do
PPG_HIP_LIB=$L/libppg_hip_$W.so PPG_BLOCKS=4096 timeout 300 $B > gpurun_out/s4/$W.json 2>/dev/null; tail -c 900 gpurun_out/s4/$W.json
done
PPG_BLOCKS=4096 timeout 300 $B > gpurun_out/s4/w4.json 2>/dev/null; tail -c 900 gpurun_out/s4/w4.json
PPG_BLOCKS=8192 timeout 300 $B --no-roofline > gpurun_out/s4/b8192.json 2>/dev/null; tail -c 300 gpurun_out/s4/b8192.json
PPG_HIP_LIB=$L/libppg_hip_w2.so PPG_BLOCKS=8192 timeout 300 $B --no-roofline > gpurun_out/s4/w2_b8192.json 2>/dev/null; tail -c 300 gpurun_out/s4/w2_b8192.json
