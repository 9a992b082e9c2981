#!/bin/bash
# GPU box, round 3, session 15: how many live lanes a wave may carry and still trace cooperatively (PPG_COOP_MAX builds)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s15
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
L=$R/practical-path-guiding_amd/lib
for rep in 1 2; do
  $B > $OUT/c6_$rep.json 2>> $OUT/err.log
  for c in 3 10 16 24; do PPG_HIP_LIB=$L/libppg_hip_c$c.so $B > $OUT/c${c}_$rep.json 2>> $OUT/err.log; done
done
for c in 10 16 24; do PPG_HIP_LIB=$L/libppg_hip_c$c.so python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/c${c}_127.json 2>> $OUT/err.log; done
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/c6_127.json 2>> $OUT/err.log
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s15.//'
