#!/bin/bash
# one short call: the optimiser-dependent parity tests with the new k_adam_apply, then 127 passes with it and with the build before it
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_adam; mkdir -p $OUT
cd $R; export TMPDIR=/tmp; ulimit -c 0
timeout 60 python -m pytest tests -m gpu -x -q -k "golden or learned_fraction or room_stand_in or (stepwise and not 64)" 2>&1 | tail -2
cd /tmp
B="timeout 40 python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call --repeats 3"
$B > $OUT/new.json 2>> $OUT/err.log
PPG_HIP_LIB=$R/practical-path-guiding_amd/lib/libppg_hip_base.so $B > $OUT/base.json 2>> $OUT/err.log
for f in new base; do python -c "
import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', round(d['value'],1), [round(x,1) for x in d['repeats']['values']], d['roofline']['kernels_ms'].get('k_adam_apply'))"; done
