#!/bin/bash
# GPU box, round 6, session 6: the final tree — the whole GPU suite first, then the evidence for profiles/r06_* (tools/r06_final.sh), the probes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_s6
mkdir -p $OUT $R/gpurun_out/profiles
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $OUT/tests_all.log 2>&1
tail -3 $OUT/tests_all.log | tee $R/gpurun_out/profiles/r06_gpu_suite_final.txt
tools/r06_final.sh 2>&1 | tail -12
cd /tmp
python $R/tools/dist_overhead_probe.py 20 2>&1 | grep "plain" | tee $OUT/dist_overhead.txt
