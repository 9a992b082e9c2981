#!/bin/bash
# GPU box, round 3, session 23: path state carried in registers through the tail, A/B on ONE box
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s23
mkdir -p $OUT
cd $R && timeout 600 python -m pytest tests/test_real_scenes.py tests/test_gpu_parity.py -m gpu -x -q -k "real or kitchen or room or null or envmap or tuning or nee or sphere or unbounded or improved" 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
L=$R/practical-path-guiding_amd/lib
$B > $OUT/warm.json 2>> $OUT/err.log
for rep in 1 2 3; do
  $B > $OUT/new_$rep.json 2>> $OUT/err.log
  PPG_HIP_LIB=$L/libppg_hip_old.so $B > $OUT/old_$rep.json 2>> $OUT/err.log
done
for v in new old; do
  l=$L/libppg_hip_$v.so; [ $v = new ] && l=$L/libppg_hip.so
  PPG_HIP_LIB=$l python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/${v}_127.json 2>> $OUT/err.log
  PPG_HIP_LIB=$l python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call > $OUT/${v}_timing.json 2>> $OUT/err.log
done
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s23.//'
python - <<P
import json
for v in ('new','old'):
    d=json.load(open('$OUT/%s_timing.json'%v)); k=d['roofline']['kernels_ms']; print(v, {n:round(k[n],2) for n in ('k_tail','k_trace','k_shade<full>','k_commit')})
P
