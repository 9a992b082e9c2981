#!/bin/bash
# GPU box: alternate environment variants of the driver's command on ONE box.  usage: ab_env.sh <out-tag> <reps> "VAR=1 VAR2=x" "..." ...  ("-" = no variables)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; REPS=$2; shift 2
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
$B > $OUT/warm.json 2>> $OUT/err.log
for rep in $(seq 1 $REPS); do
  k=0
  for v in "$@"; do
    k=$((k+1))
    if [ "$v" = "-" ]; then $B > $OUT/v${k}_$rep.json 2>> $OUT/err.log; else env $v $B > $OUT/v${k}_$rep.json 2>> $OUT/err.log; fi
  done
done
k=0
for v in "$@"; do k=$((k+1)); echo "v$k [$v]: $(grep -h -o '"value": [0-9.]*' $OUT/v${k}_*.json | sed 's/"value": //' | tr '\n' ' ')"; done
