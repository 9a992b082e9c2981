#!/bin/bash
# GPU box, round 3, session 10: the driver's command as is + the rocprofv3 evidence for profiles/r03_*
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s10
mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/profiles/r03_bench_default_plain.json 2> $OUT/plain.err
python $R/tools/collect_profiles_r03.py stats traffic wait > $OUT/collect.log 2>&1
python $R/bench.py --steps 127 --warmup 5 --no-rmse > $R/gpurun_out/profiles/r03_bench_127_passes.json 2>> $OUT/err.log
tail -30 $OUT/collect.log
head -c 600 $R/gpurun_out/profiles/r03_bench_default_plain.json
