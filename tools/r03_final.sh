#!/bin/bash
# GPU box, round 3, final build: the driver's command as the box's first process, the rocprofv3 evidence for profiles/r03_*, 127 and 1023 passes,
# the batch log, then the GPU test suite and the smoke entry
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_final
mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/profiles/r03_bench_default_plain.json 2> $OUT/plain.err
python $R/tools/collect_profiles_r03.py stats traffic wait > $OUT/collect.log 2>&1
python $R/bench.py --steps 127 --warmup 5 --no-rmse > $R/gpurun_out/profiles/r03_bench_127_passes.json 2>> $OUT/err.log
python $R/bench.py --steps 1023 --warmup 5 --no-rmse --no-cpu --no-secondary > $R/gpurun_out/profiles/r03_bench_1023_passes.json 2>> $OUT/err.log
PPG_DEBUG_BATCH=1 python $R/bench.py --steps 20 --warmup 0 --no-rmse --no-cpu --no-secondary --no-roofline --no-single-call > $OUT/debug20.json 2> $R/gpurun_out/profiles/r03_batches_20_passes.log
python $R/tools/trace_timeline.py $(find $R/gpurun_out/prof_stats -name "*kernel_trace.csv" | head -1) > $OUT/timeline_default.txt 2>&1
tail -12 $OUT/collect.log
for f in r03_bench_default_plain r03_bench_127_passes r03_bench_1023_passes; do python -c "
import json,sys; d=json.load(open('$R/gpurun_out/profiles/$f.json')); print('$f', d['value'], d.get('vs_reference_log'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['kernels_ms'])"; done
cd $R && timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
cd $R && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
