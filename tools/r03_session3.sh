#!/bin/bash
# GPU box, round 3, session 3: dense wavefront — parity tests, hand-over sensitivity, sparse-regime latency of k_tail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s3
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline"
PPG_DEBUG_BATCH=1 $B > $OUT/debug.json 2> $OUT/debug.err
for min in 786432 524288 393216 262144 196608 131072 65536; do
  PPG_TAIL_MIN=$min PPG_TAIL_DIV=1000000 $B > $OUT/tailmin_$min.json 2>> $OUT/err.log
done
PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 PPG_BOUNCE_MARGIN=8 $B > $OUT/margin8.json 2>> $OUT/err.log
PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 PPG_DEBUG_BATCH=1 $B > $OUT/debug262k.json 2> $OUT/debug262k.err
PPG_BULK_BOUNCES=0 PPG_DEBUG_BATCH=1 python $R/tools/tail_latency_probe.py 8 8 31 > $OUT/lat_8x8.json 2> $OUT/lat_8x8.err
PPG_BULK_BOUNCES=0 PPG_DEBUG_BATCH=1 python $R/tools/tail_latency_probe.py 64 36 31 > $OUT/lat_64x36.json 2> $OUT/lat_64x36.err
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/plain127.json 2>> $OUT/err.log
python $R/bench.py --scene cbox --steps 63 --warmup 3 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/cbox63.json 2>> $OUT/err.log
grep -H -o '"value": [0-9.]*' $OUT/*.json | head -40
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log; tail -5 $OUT/pytest.log
