#!/bin/bash
# GPU box, round 3, session 9: full GPU suite (owner-sharded optimiser, C++ reducer, switches, real scenes) + bench sanity
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s9
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline"
for k in 1; do $B > $OUT/plain_$k.json 2>> $OUT/err.log; done
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/plain127.json 2>> $OUT/err.log
python $R/bench.py --scene cbox --steps 255 --warmup 3 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/cbox255.json 2>> $OUT/err.log
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s9.//'
cd $R && timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/pytest.log; tail -8 $OUT/pytest.log
