#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_s13; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
for v in libppg_hip.so libppg_hip_noatom.so; do
  PPG_HIP_LIB=$R/practical-path-guiding_amd/lib/$v timeout 300 python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-single-call --repeats 1 > $OUT/k127_$v.json 2>> $OUT/err.log
done
python - $OUT/k127_libppg_hip.so.json $OUT/k127_libppg_hip_noatom.so.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(f.split("/")[-1], "value %.1f" % d["value"], {k: v for k, v in r["kernels_ms"].items() if "commit" in k or "adam" in k})
PY
