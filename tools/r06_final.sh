#!/bin/bash
# GPU box, round 6, shipped build: the driver's command as the box's first process, the rocprofv3 evidence for profiles/r06_* (kernel stats of
# the driver's command, PMC traffic and wave cycles of the SAME build), 127 and 1023 passes, the batch log, other workloads
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_final
mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/profiles/r06_bench_default_plain.json 2> $OUT/plain.err
python $R/tools/collect_profiles_r06.py traffic wait > $OUT/collect.log 2>&1
python $R/tools/collect_profiles_r06.py stats >> $OUT/collect.log 2>&1
python $R/bench.py --steps 127 --warmup 5 --no-rmse > $R/gpurun_out/profiles/r06_bench_127_passes.json 2>> $OUT/err.log
python $R/bench.py --steps 1023 --warmup 5 --no-rmse --no-cpu --no-secondary --repeats 3 > $R/gpurun_out/profiles/r06_bench_1023_passes.json 2>> $OUT/err.log
PPG_DEBUG_BATCH=1 python $R/bench.py --steps 20 --warmup 0 --no-rmse --no-cpu --no-secondary --no-roofline --no-single-call --repeats 1 > $OUT/debug20.json 2> $R/gpurun_out/profiles/r06_batches_20_passes.log
tail -12 $OUT/collect.log
for f in r06_bench_default_plain r06_bench_127_passes r06_bench_1023_passes; do python -c "
import json,sys; d=json.load(open('$R/gpurun_out/profiles/$f.json')); print('$f', d['value'], d['repeats']['values'], d.get('vs_reference_log'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['kernels_ms'])"; done
B="python $R/bench.py --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call --repeats 3 --warmup 5"
$B --scene-file $R/scratch/spaceship.ppgs --size-override --width 1920 --height 1080 --steps 255 > $OUT/ship.json 2>> $OUT/err.log
$B --scene torus --steps 255 > $OUT/torus.json 2>> $OUT/err.log
$B --scene room --steps 63 > $OUT/room.json 2>> $OUT/err.log
python - $OUT/ship.json $OUT/torus.json $OUT/room.json > $R/gpurun_out/profiles/r06_other_workloads.json <<'PY'
import json, sys
out = {}
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    out[f.split("/")[-1][:-5]] = {"workload": d["config"]["workload"], "value": d["value"], "values": d["repeats"]["values"], "unit": d["unit"]}
print(json.dumps(out, indent=1))
PY
cat $R/gpurun_out/profiles/r06_other_workloads.json | head -20
