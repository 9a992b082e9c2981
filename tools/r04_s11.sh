#!/bin/bash
# full GPU test suite + the driver's command (without the CPU baseline leg)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_s11; mkdir -p $OUT
cd $R; ulimit -c 0; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
cd /tmp
( time timeout 600 python $R/bench.py --steps 20 --warmup 5 --no-cpu > $OUT/bench20.json 2> $OUT/bench20.err ) 2>&1 | grep real
python - $OUT/bench20.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], d["repeats"], "single_call", d.get("single_call", {}).get("value"))
r = d["roofline"]; print(r["kernel"], r["frac"], r["kernels_ms"], r.get("tail_critical_path", {}).get("us_per_bounce_of_the_longest_path"))
t = d.get("time_to_rmse", {}); print({k: t.get(k) for k in ("seconds_to_mape", "spp_to_mape", "seeds_meeting_target", "all_seeds_meet_target")})
print(d["config"]["scene_file"], d.get("secondary"))
PY
