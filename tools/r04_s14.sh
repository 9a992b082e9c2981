#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_s14; mkdir -p $OUT
export PPG_AB_TESTS="kitchen_improved_against_oracle"
bash $R/tools/ab.sh r04_s14a 2 127 "-|" "-|PPG_SORT_RAYS=2" "-|PPG_SORT_RAYS=64"
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
for v in 0 2 64; do
  PPG_SORT_RAYS=$v timeout 300 python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call --repeats 1 > $OUT/k127_$v.json 2>> $OUT/err.log
done
python - $OUT/k127_0.json $OUT/k127_2.json $OUT/k127_64.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f.split("/")[-1], {k: v for k, v in r["kernels_ms"].items() if "trace" in k or "sort" in k or "shade" in k}, r["operation_counts"].get("bvh4_nodes_per_ray"), r["operation_counts"].get("triangles_tested_per_ray"))
PY
