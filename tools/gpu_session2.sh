#!/bin/bash
# round-2 GPU session 2: full parity suite (textures, bump maps, rounds), then the KITCHEN headline bench with all blocks and its rocprof summary
set -x
mkdir -p gpurun_out/s2
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/s2/pytest.log
tail -5 gpurun_out/s2/pytest.log
timeout 900 python bench.py --steps 127 --warmup 5 > gpurun_out/s2/bench_kitchen127.json 2> gpurun_out/s2/bench_kitchen127.err; tail -c 6000 gpurun_out/s2/bench_kitchen127.json; tail -5 gpurun_out/s2/bench_kitchen127.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-rmse --no-secondary > gpurun_out/s2/bench_kitchen20.json 2> gpurun_out/s2/bench_kitchen20.err; tail -c 3000 gpurun_out/s2/bench_kitchen20.json
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s2/prof_k20 -o k20 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/s2/prof_k20.log 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/s2/prof_k20 -name "*kernel_stats*" | head; find gpurun_out/s2/prof_k20 -name "*.csv" -size +3000k -delete; find gpurun_out/s2/prof_k20 -name "*.db" -delete
