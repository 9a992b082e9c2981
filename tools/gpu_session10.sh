#!/bin/bash
set -x
mkdir -p gpurun_out/s10
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s10/pytest.log 2>&1
tail -5 gpurun_out/s10/pytest.log
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
for TB in 256 512 1024 2048; do
PPG_TAIL_BLOCKS=$TB timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s10/k127_tb$TB.json 2>/dev/null
PPG_TAIL_BLOCKS=$TB timeout 300 python bench.py --steps 20 --warmup 5 $Q > gpurun_out/s10/k20_tb$TB.json 2>/dev/null
done
PPG_NO_OVERLAP=1 timeout 300 python bench.py --steps 127 --warmup 5 $Q > gpurun_out/s10/k127_serial.json 2>/dev/null
timeout 300 python bench.py --force-dist --steps 20 --warmup 5 $Q > gpurun_out/s10/bench_force_dist.json 2> gpurun_out/s10/bench_force_dist.log
cd /tmp
timeout 1500 python $GRAFT_REPO_ROOT/tools/collect_profiles_r02.py stats traffic wait > $GRAFT_REPO_ROOT/gpurun_out/s10/collect.log 2>&1
tail -40 $GRAFT_REPO_ROOT/gpurun_out/s10/collect.log
