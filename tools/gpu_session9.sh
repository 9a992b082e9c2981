#!/bin/bash
set -x
mkdir -p gpurun_out/s9
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s9/pytest.log 2>&1
tail -5 gpurun_out/s9/pytest.log
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
for i in 1 2; do
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s9/k127_overlap_$i.json 2>/dev/null
PPG_NO_OVERLAP=1 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s9/k127_serial_$i.json 2>/dev/null
done
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s9/k20_overlap.json 2>/dev/null
PPG_NO_OVERLAP=1 timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s9/k20_serial.json 2>/dev/null
timeout 300 python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --all-diffuse > gpurun_out/s9/k127_all_diffuse.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s9/bench_default.json 2> gpurun_out/s9/bench_default.err
tail -c 600 gpurun_out/s9/bench_default.err
