#!/bin/bash
# GPU box, round 3, session 2: per-batch live counts, fixed numbers of bulk bounces, sparse-regime latency of k_tail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s2
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline"
PPG_DEBUG_BATCH=1 $B > $OUT/debug.json 2> $OUT/debug.err
for k in 0 1 2 3 4 6 9 12 16 24; do
  PPG_BULK_BOUNCES=$k $B > $OUT/bulk_$k.json 2>> $OUT/err.log
done
PPG_BULK_BOUNCES=0 PPG_DEBUG_BATCH=1 python $R/tools/tail_latency_probe.py 8 8 31 > $OUT/lat_8x8.json 2> $OUT/lat_8x8.err
PPG_BULK_BOUNCES=0 PPG_DEBUG_BATCH=1 python $R/tools/tail_latency_probe.py 64 36 31 > $OUT/lat_64x36.json 2> $OUT/lat_64x36.err
PPG_BULK_BOUNCES=0 PPG_DEBUG_BATCH=1 python $R/tools/tail_latency_probe.py 320 180 31 > $OUT/lat_320x180.json 2> $OUT/lat_320x180.err
grep -H -o '"value": [0-9.]*' $OUT/*.json | head -40
