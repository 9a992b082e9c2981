#!/bin/bash
# GPU box, round 3, session 22: traversal variants on ONE box (axis select, grouped pushes, leaf vote, leaf size)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s22
mkdir -p $OUT
cd $R && timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kitchen or large_scene or room or sphere or null" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
L=$R/practical-path-guiding_amd/lib
$B > $OUT/warm.json 2>> $OUT/err.log
for rep in 1 2; do
  $B > $OUT/base_$rep.json 2>> $OUT/err.log
  for v in nosel nopush vote8 vote24 vote32; do PPG_HIP_LIB=$L/libppg_hip_$v.so $B > $OUT/${v}_$rep.json 2>> $OUT/err.log; done
  for l in 2 3 6 8; do PPG_BVH_LEAF=$l $B > $OUT/leaf${l}_$rep.json 2>> $OUT/err.log; done
done
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s22.//'
