#!/bin/bash
# GPU box, round 3, session 4: smaller FULL kernels (merged BSDF call sites, out-of-line transcendentals)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s4
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline"
PPG_DEBUG_BATCH=1 $B > $OUT/debug.json 2> $OUT/debug.err
$B > $OUT/plain.json 2>> $OUT/err.log
for min in 524288 262144 131072; do
  PPG_TAIL_MIN=$min PPG_TAIL_DIV=1000000 $B > $OUT/tailmin_$min.json 2>> $OUT/err.log
done
PPG_BULK_BOUNCES=0 PPG_DEBUG_BATCH=1 python $R/tools/tail_latency_probe.py 8 8 31 > $OUT/lat_8x8.json 2> $OUT/lat_8x8.err
PPG_BULK_BOUNCES=0 PPG_DEBUG_BATCH=1 python $R/tools/tail_latency_probe.py 64 36 31 > $OUT/lat_64x36.json 2> $OUT/lat_64x36.err
python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary > $OUT/timing.json 2>> $OUT/err.log
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/plain127.json 2>> $OUT/err.log
grep -H -o '"value": [0-9.]*' $OUT/*.json | head -40
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log; tail -5 $OUT/pytest.log
