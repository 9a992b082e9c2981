#!/bin/bash
set -x
mkdir -p gpurun_out/s20
export TMPDIR=/tmp
timeout 300 python tools/phase_timing.py kitchen 20 > gpurun_out/s20/phase_k20.log 2>&1
timeout 300 python tools/phase_timing.py kitchen 127 > gpurun_out/s20/phase_k127.log 2>&1
timeout 300 python tools/phase_timing.py kitchen 20 8 > gpurun_out/s20/phase_k20_w8.log 2>&1
timeout 300 python tools/phase_timing.py kitchen 127 8 > gpurun_out/s20/phase_k127_w8.log 2>&1
tail -3 gpurun_out/s20/*.log
