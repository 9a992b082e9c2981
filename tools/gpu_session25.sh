#!/bin/bash
set -x
mkdir -p gpurun_out/s25
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s25/pytest.log 2>&1
tail -4 gpurun_out/s25/pytest.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/s25/smoke.log 2>&1
tail -2 gpurun_out/s25/smoke.log
cd /tmp
timeout 1500 python $GRAFT_REPO_ROOT/tools/collect_profiles_r02.py stats traffic wait > $GRAFT_REPO_ROOT/gpurun_out/s25/collect.log 2>&1
tail -12 $GRAFT_REPO_ROOT/gpurun_out/s25/collect.log
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/s25/bench_default.json 2> gpurun_out/s25/bench_default.err
timeout 300 python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s25/k127_kernels.json 2>/dev/null
