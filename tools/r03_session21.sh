#!/bin/bash
# GPU box, round 3, session 21: do independent renders overlap on one GPU? (tile shards in concurrent contexts)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s21
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 300 python $R/tools/overlap_probe.py 20 1 2 3 4 2>&1 | tee $OUT/probe20.txt | tail -12
timeout 300 python $R/tools/overlap_probe.py 127 1 2 4 2>&1 | tee $OUT/probe127.txt | tail -8
