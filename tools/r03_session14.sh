#!/bin/bash
# GPU box, round 3, session 14: wave-cooperative traversal in sparse tail waves
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s14
mkdir -p $OUT
cd $R && timeout 900 python -m pytest tests/test_real_scenes.py tests/test_gpu_parity.py -m gpu -x -q -k "real or room or tuning or kitchen or spaceship or torus or null or envmap or sphere or large_scene" 2>&1 | tail -5
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
for k in 1 2 3; do $B > $OUT/plain_$k.json 2>> $OUT/err.log; done
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/plain127.json 2>> $OUT/err.log
python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call > $OUT/timing.json 2>> $OUT/err.log
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s14.//'
python -c "
import json; d=json.load(open('$OUT/timing.json')); print(d['roofline']['kernels_ms']); print(d['roofline'].get('tail_critical_path'))"
