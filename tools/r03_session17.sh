#!/bin/bash
# GPU box, round 3, session 17: A/B of the early S-tree lookup in shade_one on ONE box
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s17
mkdir -p $OUT
cd $R && timeout 900 python -m pytest tests/test_real_scenes.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
L=$R/practical-path-guiding_amd/lib
for rep in 1 2 3; do
  $B > $OUT/spf_$rep.json 2>> $OUT/err.log
  PPG_HIP_LIB=$L/libppg_hip_nospf.so $B > $OUT/nospf_$rep.json 2>> $OUT/err.log
done
for v in spf nospf; do
  l=$L/libppg_hip_$v.so; [ $v = spf ] && l=$L/libppg_hip.so
  PPG_HIP_LIB=$l python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/${v}_127.json 2>> $OUT/err.log
  PPG_HIP_LIB=$l python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call > $OUT/${v}_timing.json 2>> $OUT/err.log
done
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s17.//'
python - <<P
import json
for v in ('spf','nospf'):
    d=json.load(open('$OUT/%s_timing.json'%v)); k=d['roofline']['kernels_ms']; print(v, {n:round(k[n],2) for n in ('k_tail','k_trace','k_shade<full>','k_commit')})
P
