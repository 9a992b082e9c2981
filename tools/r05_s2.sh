#!/bin/bash
# round 5, session 2: path-major k_commit_records, k_splat_sorted beside the optimiser; bench.py's new legs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q -k "tuning or golden or learned_fraction or round_hook or room_stand_in or stepwise or mask or kitchen_improved or cancel or cpp_ or rccl or sharded_contexts" 2>&1 | tail -5
export PPG_AB_KERNELS=1
bash tools/ab.sh r05_s2_20 1 20 "-|" "-|PPG_NO_SORTED_COMMIT=1 PPG_ADAM_UNORDERED=1"
bash tools/ab.sh r05_s2_127 1 127 "-|" "-|PPG_NO_SORTED_COMMIT=1 PPG_ADAM_UNORDERED=1"
bash tools/ab.sh r05_s2_1023 1 1023 "-|"
cd /tmp; timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-rmse --no-secondary > $R/gpurun_out/r05_s2_bench20.json 2> $R/gpurun_out/r05_s2_bench20.err; tail -c 1500 $R/gpurun_out/r05_s2_bench20.err
python - <<'PY'
import json
d=json.loads(open('/root/repo/gpurun_out/r05_s2_bench20.json').read().strip().splitlines()[-1])
print(d["value"], d["repeats"]["values"], d.get("cpu_baseline"), d["data"])
print(json.dumps(d["roofline"]["per_kernel"], indent=0)[:3000])
print(d["roofline"]["operation_counts"])
PY
