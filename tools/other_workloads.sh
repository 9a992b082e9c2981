#!/bin/bash
# GPU box: the other workloads of DESIGN.md section 7 with the current build (SPACESHIP 1080p, torus stand-in, room stand-in) -> gpurun_out/profiles/r03_other_workloads.json
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/other
mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp; export TMPDIR=/tmp
F="--warmup 3 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
python $R/bench.py --scene-file $R/scratch/spaceship.ppgs --steps 255 $F --size-override --width 1920 --height 1080 > $OUT/spaceship1080.json 2>> $OUT/err.log
python $R/bench.py --scene torus --steps 255 $F > $OUT/torus.json 2>> $OUT/err.log
python $R/bench.py --scene room --steps 63 $F > $OUT/room.json 2>> $OUT/err.log
python - <<P
import json
out={}
for n in ('spaceship1080','torus','room'):
    d=json.loads(open('$OUT/%s.json'%n).read().strip().splitlines()[-1])
    out[n]={'msamples_per_s':d['value'],'passes':d['steps'],'workload':d['config']['workload']}
    print(n, d['value'])
json.dump(out, open('$R/gpurun_out/profiles/r03_other_workloads.json','w'), indent=1)
P
