#!/bin/bash
# GPU box, round 4, after bench.py's last change (garbage collection before each timed render): the driver's command again, plain and under rocprofv3 --kernel-trace --stats
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04_final; mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/profiles/r04_bench_default_plain.json 2> $OUT/plain.err
python $R/tools/collect_profiles_r04.py stats > $OUT/collect3.log 2>&1
tail -5 $OUT/collect3.log
for f in r04_bench_default_plain r04_bench_default; do python -c "
import json,sys; d=json.load(open('$R/gpurun_out/profiles/$f.json')); print('$f', d['value'], d['repeats']['values'], d.get('vs_reference_log'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d.get('single_call',{}).get('value'))"; done
