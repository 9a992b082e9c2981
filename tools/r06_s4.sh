#!/bin/bash
# GPU box, round 6, session 4: after the flush fix — the stragglers' tests, then the driver's command / 127 / 1023 passes against PPG_SPLIT_DEPTH=0
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_s4
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "stragglers or tuning_switches or two_ranks or full_size" > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
B="timeout 600 python $R/bench.py --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
$B --steps 20 > $OUT/b20.json 2> $OUT/err.log
PPG_FINAL_HALVES=1 $B --steps 20 > $OUT/b20_halves.json 2>> $OUT/err.log
$B --steps 20 > $OUT/b20_b.json 2>> $OUT/err.log
$B --steps 127 --repeats 3 > $OUT/b127.json 2>> $OUT/err.log
PPG_SPLIT_DEPTH=0 $B --steps 127 --repeats 3 > $OUT/b127_nosplit.json 2>> $OUT/err.log
$B --steps 1023 --repeats 3 > $OUT/b1023.json 2>> $OUT/err.log
PPG_SPLIT_DEPTH=0 $B --steps 1023 --repeats 3 > $OUT/b1023_nosplit.json 2>> $OUT/err.log
for f in b20 b20_halves b20_b b127 b127_nosplit b1023 b1023_nosplit; do python -c "
import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), [round(v,1) for v in d['repeats']['values']])"; done
