#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_s7; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for v in libppg_hip_r03.so libppg_hip.so; do
  PPG_HIP_LIB=$R/practical-path-guiding_amd/lib/$v python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call > $OUT/k20_$v.json 2>> $OUT/err.log
  PPG_HIP_LIB=$R/practical-path-guiding_amd/lib/$v python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call --repeats 1 > $OUT/k127_$v.json 2>> $OUT/err.log
done
python - $OUT/k20_libppg_hip_r03.so.json $OUT/k20_libppg_hip.so.json $OUT/k127_libppg_hip_r03.so.json $OUT/k127_libppg_hip.so.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f.split("/")[-1], "value %.1f" % d["value"], {k: v for k, v in r["kernels_ms"].items()}, "tail_cp", r.get("tail_critical_path", {}).get("us_per_bounce_of_the_longest_path"))
PY
