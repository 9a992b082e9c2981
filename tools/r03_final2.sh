#!/bin/bash
# GPU box, round 3, final build (leaves of <= 3 triangles): the driver's command as the box's first process, the GPU test suite, the smoke entry,
# then the kernel trace of the driver's command and the 127 / 1023-pass lines
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r03_final2
mkdir -p $OUT $R/gpurun_out/profiles
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/profiles/r03_bench_default_plain.json 2> $OUT/plain.err
python -c "
import json; d=json.load(open('$R/gpurun_out/profiles/r03_bench_default_plain.json')); print('plain', d['value'], d['single_call']['value'], d['roofline']['kernels_ms'])"
cd $R && timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_real_scenes.py -m gpu -q -n 3 2>&1 | tail -3
cd $R && timeout 600 python -m pytest tests -m gpu -x -q --ignore=tests/test_gpu_parity.py --ignore=tests/test_real_scenes.py 2>&1 | tail -3
cd $R && python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
cd /tmp
python $R/tools/collect_profiles_r03.py stats > $OUT/collect.log 2>&1
python $R/bench.py --steps 127 --warmup 5 --no-rmse --no-cpu --no-secondary > $R/gpurun_out/profiles/r03_bench_127_passes.json 2>> $OUT/err.log
python $R/bench.py --steps 1023 --warmup 5 --no-rmse --no-cpu --no-secondary > $R/gpurun_out/profiles/r03_bench_1023_passes.json 2>> $OUT/err.log
PPG_DEBUG_BATCH=1 python $R/bench.py --steps 20 --warmup 0 --no-rmse --no-cpu --no-secondary --no-roofline --no-single-call > $OUT/debug20.json 2> $R/gpurun_out/profiles/r03_batches_20_passes.log
for f in r03_bench_default r03_bench_127_passes r03_bench_1023_passes; do python -c "
import json,sys; d=json.load(open('$R/gpurun_out/profiles/$f.json')); print('$f', d['value'], d.get('single_call',{}).get('value'), d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['kernels_ms'])"; done
