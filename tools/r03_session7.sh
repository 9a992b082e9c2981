#!/bin/bash
# GPU box, round 3, session 7: which of the tail's switches pay (main-stream tail with k_commit beside it = default)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s7
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline"
run() { name=$1; shift; env "$@" $B > $OUT/$name.json 2>> $OUT/err.log; }
run plain A=1
run plain2 A=1
run nooverlap PPG_NO_OVERLAP=1
run setprio PPG_TAIL_SETPRIO=1
run b4096 PPG_TAIL_BLOCKS=4096
run b768 PPG_TAIL_BLOCKS=768
run b2048 PPG_TAIL_BLOCKS=2048
run min524k PPG_TAIL_MIN=524288
run min1M PPG_TAIL_MIN=1048576
run div6 PPG_TAIL_DIV=6
run div24 PPG_TAIL_DIV=24
run side PPG_SIDE_TAILS=1
run side_prio PPG_SIDE_TAILS=1 PPG_TAIL_PRIO=1
run side_sub4M PPG_SIDE_TAILS=1 PPG_SUB_PATHS=4000000
run side_sub6M PPG_SIDE_TAILS=1 PPG_SUB_PATHS=6000000
run side_sub6M_b512 PPG_SIDE_TAILS=1 PPG_SUB_PATHS=6000000 PPG_TAIL_BLOCKS=512
run gens2_64 PPG_TAIL_GENS=2 PPG_TAIL_GEN=64
run margin1 PPG_BOUNCE_MARGIN=1
run margin6 PPG_BOUNCE_MARGIN=6
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/plain127.json 2>> $OUT/err.log
PPG_DEBUG_BATCH=1 $B > $OUT/debug.json 2> $OUT/debug.err
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s7.//'
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log; tail -5 $OUT/pytest.log
