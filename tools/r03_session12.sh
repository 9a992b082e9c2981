#!/bin/bash
# GPU box, round 3, session 12: the cleaned-up build — full GPU suite + bench sanity
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_s12
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
for k in 1 2 3; do $B > $OUT/plain_$k.json 2>> $OUT/err.log; done
python $R/bench.py --steps 127 --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/plain127.json 2>> $OUT/err.log
python $R/bench.py --scene-file $R/scratch/spaceship.ppgs --steps 255 --warmup 3 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call --size-override --width 1920 --height 1080 > $OUT/spaceship1080.json 2>> $OUT/err.log
python $R/bench.py --scene torus --steps 255 --warmup 3 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/torus.json 2>> $OUT/err.log
python $R/bench.py --scene room --steps 63 --warmup 3 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call > $OUT/room.json 2>> $OUT/err.log
grep -H -o '"value": [0-9.]*' $OUT/*.json | sed 's/.*r03_s12.//'
cd $R && timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest.log; tail -5 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
