#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_s17; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
B="timeout 300 python $R/bench.py --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call --repeats 3 --warmup 5"
for rep in 1 2; do
for L in soa aos pack; do
  PPG_PATH_LAYOUT=$L $B --scene-file $R/scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 > $OUT/ship_${L}_$rep.json 2>> $OUT/err.log
  PPG_PATH_LAYOUT=$L $B --scene cbox --steps 63 > $OUT/cbox_${L}_$rep.json 2>> $OUT/err.log
  PPG_PATH_LAYOUT=$L $B --scene torus --steps 63 > $OUT/torus_${L}_$rep.json 2>> $OUT/err.log
  PPG_PATH_LAYOUT=$L $B --steps 1023 --repeats 1 > $OUT/k1023_${L}_$rep.json 2>> $OUT/err.log
done; done
for f in $OUT/*.json; do echo "$(basename $f) $(python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(round(d['value'],1), [round(x,1) for x in d['repeats']['values']])")"; done
