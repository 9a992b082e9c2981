#!/bin/bash
# round 5, session 7: the whole GPU suite on the final build (guarded: one quick render first, per-test and overall timeouts), then k_sort_slices
# with the per-triangle class table against without (PPG_NO_TRI_CLASS)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 200 python -m pytest tests -m gpu -x -q -k "kitchen_improved_against_oracle" --timeout 150 2>&1 | tail -3
if [ ${PIPESTATUS[0]} -ne 0 ]; then echo "SANITY FAILED: stopping"; exit 1; fi
timeout 700 python -m pytest tests -m gpu -x -q --timeout 200 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
B="timeout 150 python $R/bench.py --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call"
for v in "" "PPG_NO_TRI_CLASS=1" "" "PPG_NO_TRI_CLASS=1"; do
  env $v $B --steps 127 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] 127 passes:', round(d['value'],1), [round(x,1) for x in d['repeats']['values']], 'sort_slices', d['roofline']['kernels_ms'].get('k_sort_slices'))"
done
for v in "" "PPG_NO_TRI_CLASS=1"; do
  env $v $B --steps 1023 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] 1023 passes:', round(d['value'],1), [round(x,1) for x in d['repeats']['values']], 'sort_slices', d['roofline']['kernels_ms'].get('k_sort_slices'))"
done
