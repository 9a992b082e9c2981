#!/bin/bash
# GPU box, round 3, session 8: BVH prefetch A/B, PMC view of the sparse tail, the new real-scene tests
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s8
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline"
PF=$R/practical-path-guiding_amd/lib/libppg_hip_pf.so
for k in 1 2 3; do
  $B > $OUT/plain_$k.json 2>> $OUT/err.log
  PPG_HIP_LIB=$PF $B > $OUT/pf_$k.json 2>> $OUT/err.log
done
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/plain127.json 2>> $OUT/err.log
PPG_HIP_LIB=$PF python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline > $OUT/pf127.json 2>> $OUT/err.log
python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary > $OUT/timing.json 2>> $OUT/err.log
PPG_HIP_LIB=$PF python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary > $OUT/timing_pf.json 2>> $OUT/err.log
SQ="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
PPG_BULK_BOUNCES=0 rocprofv3 --pmc $SQ --output-format csv -d $OUT/pmc_sparse -o p -- python $R/tools/tail_latency_probe.py 8 8 31 > $OUT/pmc_sparse.json 2> $OUT/pmc_sparse.err
PPG_BULK_BOUNCES=0 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_IFETCH SQ_WAVES --output-format csv -d $OUT/pmc_sparse2 -o p -- python $R/tools/tail_latency_probe.py 8 8 31 > $OUT/pmc_sparse2.json 2> $OUT/pmc_sparse2.err
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s8.//'
cd $R && timeout 600 python -m pytest tests/test_real_scenes.py -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest.log; tail -5 $OUT/pytest.log
