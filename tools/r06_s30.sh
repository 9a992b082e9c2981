#!/bin/bash
# GPU box: kernel-trace timeline of one 20-pass render of the final build (streams, overlaps, idle time between kernels)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/profiles $R/gpurun_out/r06_s30
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call --repeats 2 > $R/gpurun_out/r06_s30/bench.json 2> $R/gpurun_out/r06_s30/err.log
F=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_timeline.py $F > $R/gpurun_out/r06_s30/timeline_full.txt 2>&1
grep -E "k_tail|k_commit_records|k_generate|k_trace|k_splat_sorted|k_adam_apply|idle|total|k_shade" $R/gpurun_out/r06_s30/timeline_full.txt | head -150 > $R/gpurun_out/profiles/r06_final_timeline.txt
tail -25 $R/gpurun_out/r06_s30/timeline_full.txt
