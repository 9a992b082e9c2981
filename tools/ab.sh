#!/bin/bash
# GPU box: alternate variants (a build of the library and / or environment switches) of the driver's command on ONE box.
# usage: ab.sh <out-tag> <reps> <passes> "<lib>|<VAR=1 VAR2=x>" ...   (lib relative to practical-path-guiding_amd/lib/, "-" = the shipped libppg_hip.so; env may be empty)
# PPG_AB_TESTS="<pytest -k expression>": first run those GPU tests with every variant (parity before speed).
# PPG_AB_KERNELS=1: keep bench.py's instrumented render and print every variant's kernel times (ms) beside its value.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; REPS=$2; PASSES=$3; shift 3
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
NR=--no-roofline; [ -n "$PPG_AB_KERNELS" ] && NR=""
B="timeout 300 python $R/bench.py --steps $PASSES --warmup 5 --no-cpu --no-rmse --no-secondary $NR --no-single-call"
lib() { if [ "$1" = "-" ] || [ -z "$1" ]; then echo $R/practical-path-guiding_amd/lib/libppg_hip.so; else echo $R/practical-path-guiding_amd/lib/$1; fi; }
if [ -n "$PPG_AB_TESTS" ]; then
  for v in "$@"; do
    L=${v%%|*}; E=${v#*|}; [ "$E" = "$v" ] && E=""
    echo "== tests with [$v]"; (cd $R && env PPG_HIP_LIB=$(lib $L) $E timeout 900 python -m pytest tests -m gpu -x -q -k "$PPG_AB_TESTS" 2>&1 | tail -3)
  done
fi
$B > $OUT/warm.json 2>> $OUT/err.log
for rep in $(seq 1 $REPS); do
  k=0
  for v in "$@"; do
    k=$((k+1)); L=${v%%|*}; E=${v#*|}; [ "$E" = "$v" ] && E=""
    env PPG_HIP_LIB=$(lib $L) $E $B > $OUT/v${k}_$rep.json 2>> $OUT/err.log
  done
done
k=0
for v in "$@"; do k=$((k+1)); echo "v$k [$v] $PASSES passes: $(python - $OUT/v${k}_*.json <<'PY'
import json, sys
out = []
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        out.append("%.1f (%s)" % (d["value"], " ".join("%.1f" % x for x in d.get("repeats", {}).get("values", []))))
        km = d.get("roofline", {}).get("kernels_ms")
        if km:
            out.append("{" + " ".join("%s=%.1f" % (k.replace("k_", ""), v) for k, v in km.items() if v >= 0.05) + "}")
    except Exception as e:
        out.append("?")
print("  ".join(out))
PY
)"; done
