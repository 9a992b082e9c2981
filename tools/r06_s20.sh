#!/bin/bash
# GPU box, round 6, the final tree (pair compaction in k_trace, suspended traversals in k_tail): the whole GPU suite, the evidence for
# profiles/r06_* (tools/r06_final.sh), the reducer on a one-rank RCCL communicator, the probes of section 6
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_s20
mkdir -p $OUT $R/gpurun_out/profiles
cd $R
tools/r06_final.sh 2>&1 | tail -14
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $OUT/tests_all.log 2>&1
tail -3 $OUT/tests_all.log | tee $R/gpurun_out/profiles/r06_gpu_suite_final.txt
cd /tmp
python $R/bench.py --steps 20 --warmup 5 --force-dist --no-cpu --no-rmse --no-secondary > $R/gpurun_out/profiles/r06_bench_force_dist_one_rank.json 2>> $OUT/err.log
python $R/tools/dist_overhead_probe.py 20 2>&1 | grep "plain" | tee $OUT/dist_overhead.txt
python $R/tools/shard_scaling_probe.py 2>&1 | tail -12
