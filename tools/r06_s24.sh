#!/bin/bash
# GPU box: the round's sort over the positions that hold a record (key compaction) — parity, then A/B against the sort over all positions
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out/r06_s24
timeout 1200 python -m pytest tests -m gpu -x -q -k "golden or room or kitchen or stragglers or spaceship or tuning or full_size or two_ranks or region or optim or improved" 2>&1 | tail -4
tools/ab.sh r06_s24/ab20 3 20 "-|PPG_NO_COMPACT=1" "-|"
tools/ab.sh r06_s24/ab127 2 127 "-|PPG_NO_COMPACT=1" "-|"
tools/ab.sh r06_s24/ab1023 1 1023 "-|PPG_NO_COMPACT=1" "-|"
