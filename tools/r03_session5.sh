#!/bin/bash
# GPU box, round 3, session 5: sub-batches with side-stream tails
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s5
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline"
PPG_DEBUG_BATCH=1 $B > $OUT/debug.json 2> $OUT/debug.err
$B > $OUT/plain.json 2>> $OUT/err.log
for sp in 500000 1000000 4000000 100000000; do
  PPG_SUB_PATHS=$sp $B > $OUT/sub_$sp.json 2>> $OUT/err.log
done
for min in 524288 262144 131072; do
  PPG_TAIL_MIN=$min PPG_TAIL_DIV=1000000 $B > $OUT/tailmin_$min.json 2>> $OUT/err.log
  PPG_SUB_PATHS=1000000 PPG_TAIL_MIN=$min PPG_TAIL_DIV=1000000 $B > $OUT/sub1M_tailmin_$min.json 2>> $OUT/err.log
done
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/plain127.json 2>> $OUT/err.log
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- $B > $OUT/trace.json 2> $OUT/trace.err
grep -H -o '"value": [0-9.]*' $OUT/*.json | head -40
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log; tail -5 $OUT/pytest.log
