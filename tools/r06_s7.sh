#!/bin/bash
# GPU box, round 6, session 7: the records' sort on the side stream — parity, then A/B against the previous build of the library
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_s7
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "golden or stragglers or tuning_switches or improved or two_ranks or kitchen or full_size" > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
tools/ab.sh r06_ab6 2 20 "-|" "libppg_hip_prev.so|" 2>&1 | tail -3
tools/ab.sh r06_ab7 1 127 "-|" "libppg_hip_prev.so|" 2>&1 | tail -3
tools/ab.sh r06_ab8 1 1023 "-|" "libppg_hip_prev.so|" 2>&1 | tail -3
