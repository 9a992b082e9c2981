#!/bin/bash
# GPU box, round 4, call 1: this box's baseline for the shipped build, the cycle probe of a lone path's bounce (libppg_hip_probe.so), the
# hand-over threshold re-test
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_s1
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
$B > $OUT/warm.json 2> $OUT/warm.err
for rep in 1 2; do
  $B > $OUT/base_$rep.json 2>> $OUT/err.log
  PPG_TAIL_MIN=1048576 $B > $OUT/tm1m_$rep.json 2>> $OUT/err.log
  PPG_TAIL_MIN=524288 $B > $OUT/tm512k_$rep.json 2>> $OUT/err.log
done
PPG_HIP_LIB=$R/practical-path-guiding_amd/lib/libppg_hip_probe.so PPG_DEBUG_BATCH=1 $B > $OUT/probe20.json 2> $OUT/probe20.err
PPG_HIP_LIB=$R/practical-path-guiding_amd/lib/libppg_hip_probe.so PPG_DEBUG_BATCH=1 PPG_BULK_BOUNCES=0 python $R/tools/tail_latency_probe.py 64 36 20 > $OUT/probe_small.json 2> $OUT/probe_small.err
for f in warm base_1 base_2 tm1m_1 tm1m_2 tm512k_1 tm512k_2 probe20; do echo "$f $(grep -h -o '"value": [0-9.]*' $OUT/$f.json | head -1)"; done
grep -h "ppg probe" $OUT/probe20.err | tail -3
grep -h "ppg probe" $OUT/probe_small.err | tail -2
grep -h "ppg passes" $OUT/probe20.err | tail -8
rocm-smi --showclocks 2>/dev/null | head -20
