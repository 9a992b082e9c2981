#!/bin/bash
# kernel timeline of one 20-pass render of the current build (rocprofv3 --kernel-trace) -> gaps between kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_s9; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call --repeats 1 > $OUT/bench.json 2> $OUT/err.log
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_timeline.py $F > $OUT/timeline.txt 2>&1
tail -25 $OUT/timeline.txt
grep -c . $OUT/timeline.txt
rm -rf $OUT/trace
