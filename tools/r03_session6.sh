#!/bin/bash
# GPU box, round 3, session 6: generational tails on side streams beside the next sub-batch's wavefront
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s6
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline"
run() { name=$1; shift; env "$@" $B > $OUT/$name.json 2>> $OUT/err.log; }
PPG_DEBUG_BATCH=1 $B > $OUT/debug.json 2> $OUT/debug.err
run plain A=1
run min262k PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000
run min131k PPG_TAIL_MIN=131072 PPG_TAIL_DIV=1000000
run min131k_b256 PPG_TAIL_MIN=131072 PPG_TAIL_DIV=1000000 PPG_TAIL_BLOCKS=256
run min262k_b256 PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 PPG_TAIL_BLOCKS=256
run min262k_b1024 PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 PPG_TAIL_BLOCKS=1024
run min262k_sub1M PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 PPG_SUB_PATHS=1000000
run min262k_sub4M PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 PPG_SUB_PATHS=4000000
run min262k_nosub PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 PPG_SUB_PATHS=100000000
run min262k_g16 PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 PPG_TAIL_GEN=16 PPG_TAIL_GENS=20
run min262k_g64 PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 PPG_TAIL_GEN=64 PPG_TAIL_GENS=6
run min262k_sub500k PPG_TAIL_MIN=131072 PPG_TAIL_DIV=1000000 PPG_SUB_PATHS=500000
run nooverlap PPG_NO_OVERLAP=1
PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- $B > $OUT/trace.json 2> $OUT/trace.err
PPG_TAIL_MIN=262144 PPG_TAIL_DIV=1000000 python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/min262k_127.json 2>> $OUT/err.log
grep -H -o '"value": [0-9.]*' $OUT/*.json | head -40
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log; tail -5 $OUT/pytest.log
