#!/bin/bash
# GPU box, round 6, session 1: the driver's command on this round's box (baseline), and the survival curve of KITCHEN's paths out to 64 wavefront
# bounces (PPG_BULK_BOUNCES=64 PPG_TAIL_THRESHOLD=1: the wavefront keeps every path) — how many paths are alive after bounce D
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_s1
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
B="timeout 600 python $R/bench.py --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
$B --steps 20 > $OUT/base20.json 2> $OUT/base20.err
$B --steps 127 --repeats 3 > $OUT/base127.json 2>> $OUT/base20.err
PPG_DEBUG_BATCH=1 PPG_BULK_BOUNCES=64 PPG_TAIL_THRESHOLD=1 $B --steps 127 --warmup 0 --repeats 1 > $OUT/curve127.json 2> $OUT/curve127.log
PPG_DEBUG_BATCH=1 $B --steps 127 --warmup 0 --repeats 1 > $OUT/debug127.json 2> $OUT/debug127.log
for f in base20 base127; do python -c "
import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['repeats']['values'])"; done
grep -c batch $OUT/curve127.log
