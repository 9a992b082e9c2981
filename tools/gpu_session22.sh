#!/bin/bash
set -x
mkdir -p gpurun_out/s22
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s22/pytest.log 2>&1
tail -4 gpurun_out/s22/pytest.log
for i in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s22/k20_$i.json 2>/dev/null
timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s22/k127_$i.json 2>/dev/null
done
timeout 300 python bench.py --steps 127 --warmup 127 $Q > gpurun_out/s22/k127_warm127.json 2>/dev/null
timeout 300 python tools/phase_timing.py kitchen 20 > gpurun_out/s22/phase_k20.log 2>&1
timeout 300 python bench.py --scene cbox --steps 255 --warmup 5 $Q > gpurun_out/s22/cbox.json 2>/dev/null
timeout 300 python __graft_entry__.py smoke > gpurun_out/s22/smoke.log 2>&1
tail -2 gpurun_out/s22/smoke.log
