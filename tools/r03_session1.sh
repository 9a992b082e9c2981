#!/bin/bash
# GPU box, round 3, session 1: timeline of the driver's command with the round-2 build + sensitivity of the hand-over point
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s1
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline"
$B > $OUT/plain0.json 2> $OUT/plain0.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- $B > $OUT/trace.json 2> $OUT/trace.err
for min in 786432 393216 196608 98304 32768; do
  PPG_TAIL_MIN=$min PPG_TAIL_DIV=1000000 $B > $OUT/tailmin_$min.json 2>> $OUT/err.log
done
for tb in 768 1536; do
  PPG_TAIL_BLOCKS=$tb PPG_NO_OVERLAP=1 $B > $OUT/tailblocks_$tb.json 2>> $OUT/err.log
done
python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary > $OUT/timing.json 2>> $OUT/err.log
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/plain127.json 2>> $OUT/err.log
grep -h -o '"value": [0-9.]*' $OUT/*.json | head -40
