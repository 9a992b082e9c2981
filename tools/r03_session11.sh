#!/bin/bash
# GPU box, round 3, session 11: tail generations that SPREAD the survivors over all waves
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s11
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
run() { name=$1; shift; env "$@" $B > $OUT/$name.json 2>> $OUT/err.log; }
run plain A=1
run caps_16_16_32_64 PPG_TAIL_CAPS=16,16,32,64
run caps_8_8_16_32_64 PPG_TAIL_CAPS=8,8,16,32,64
run caps_24_48 PPG_TAIL_CAPS=24,48
run caps_32 PPG_TAIL_CAPS=32
run caps_16x6 PPG_TAIL_CAPS=16,16,16,16,16,16,32,64
run caps_b2048 PPG_TAIL_CAPS=16,16,32,64 PPG_TAIL_BLOCKS=2048
run caps_b768 PPG_TAIL_CAPS=16,16,32,64 PPG_TAIL_BLOCKS=768
run caps_min1M PPG_TAIL_CAPS=16,16,32,64 PPG_TAIL_MIN=1048576
run caps_min524k PPG_TAIL_CAPS=16,16,32,64 PPG_TAIL_MIN=524288
run caps_min262k PPG_TAIL_CAPS=16,16,32,64 PPG_TAIL_MIN=262144 PPG_TAIL_DIV=48
run plain2 A=1
PPG_TAIL_CAPS=16,16,32,64 python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/caps127.json 2>> $OUT/err.log
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/plain127.json 2>> $OUT/err.log
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s11.//'
cd $R && PPG_TAIL_CAPS=4,8 timeout 600 python -m pytest tests/test_real_scenes.py tests/test_gpu_parity.py -m gpu -x -q -k "real or room or tuning or kitchen" 2>&1 | tail -5
