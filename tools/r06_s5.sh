#!/bin/bash
# GPU box, round 6, session 5: the optimiser's lanes — parity (every test with a learned fraction), then A/B against PPG_ADAM_NO_LANES
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_s5
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q -k "golden or stragglers or tuning_switches or improved or adam or optim or two_ranks or kitchen" > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
PPG_AB_KERNELS=1 tools/ab.sh r06_ab3 2 20 "-|" "-|PPG_ADAM_NO_LANES=1" "-|PPG_ADAM_LIGHT=1024" "-|PPG_ADAM_LIGHT=16384" 2>&1 | tail -5
PPG_AB_KERNELS=1 tools/ab.sh r06_ab4 1 127 "-|" "-|PPG_ADAM_NO_LANES=1" "-|PPG_ADAM_LIGHT=1024" "-|PPG_ADAM_LIGHT=16384" 2>&1 | tail -5
PPG_AB_KERNELS=1 tools/ab.sh r06_ab5 1 1023 "-|" "-|PPG_ADAM_NO_LANES=1" 2>&1 | tail -3
cd /tmp; python $R/tools/dist_overhead_probe.py 20 2>&1 | grep "plain"
