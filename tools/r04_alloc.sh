#!/bin/bash
# which render of a process asks the driver for memory (PPG_DEBUG_ALLOC): seven 20-pass renders, each timed, with the allocation log between them
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_alloc; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
PPG_DEBUG_ALLOC=1 timeout 300 python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call --repeats 7 > $OUT/bench.json 2> $OUT/alloc.log
grep -c "hipMalloc" $OUT/alloc.log; grep -v "ok (cache holds 0 MiB)" $OUT/alloc.log | head -40
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['repeats']['values'])"
