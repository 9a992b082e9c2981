#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_final; mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
python $R/tools/collect_profiles_r04.py traffic wait > $OUT/collect2.log 2>&1
tail -40 $OUT/collect2.log
