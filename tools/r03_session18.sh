#!/bin/bash
# GPU box, round 3, session 18: timeline of the driver's command (where are the gaps between kernels?)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s18
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
$B > $OUT/plain_1.json 2>> $OUT/err.log
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- $B > $OUT/trace.json 2> $OUT/trace.err
python $R/tools/trace_timeline.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/timeline.txt
grep -c . $OUT/timeline.txt
awk '$4+0 > 0.08 || /^#/' $OUT/timeline.txt | head -150
