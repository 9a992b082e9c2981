#!/bin/bash
set -x
mkdir -p gpurun_out/s26
export TMPDIR=/tmp
cd /tmp
timeout 1200 python $GRAFT_REPO_ROOT/tools/collect_profiles_r02.py stats > $GRAFT_REPO_ROOT/gpurun_out/s26/collect.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/s26/collect.log
cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary > gpurun_out/s26/k127_kernels.json 2>/dev/null
