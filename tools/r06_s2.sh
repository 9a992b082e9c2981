#!/bin/bash
# GPU box, round 6, session 2: stragglers — parity first (the new tests, then the whole GPU suite), then the driver's command, 127 and 1023 passes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_s2
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stragglers or tuning_switches or room_stand_in or golden" > $OUT/tests_new.log 2>&1
tail -15 $OUT/tests_new.log
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
B="timeout 600 python $R/bench.py --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
$B --steps 20 > $OUT/b20.json 2> $OUT/b20.err
PPG_SPLIT_DEPTH=0 $B --steps 20 > $OUT/b20_nosplit.json 2>> $OUT/b20.err
$B --steps 127 --repeats 3 > $OUT/b127.json 2>> $OUT/b20.err
$B --steps 1023 --repeats 3 > $OUT/b1023.json 2>> $OUT/b20.err
PPG_DEBUG_BATCH=1 $B --steps 20 --warmup 0 --repeats 1 > $OUT/debug20.json 2> $OUT/debug20.log
for f in b20 b20_nosplit b127 b1023; do python -c "
import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['repeats']['values'])"; done
tail -5 $OUT/b20.err
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/tests_all.log 2>&1
tail -5 $OUT/tests_all.log
