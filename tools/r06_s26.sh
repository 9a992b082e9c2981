#!/bin/bash
# GPU box: the three bench lines of the final tree once more, with bench.py counting the slice sort's bytes for k_trace
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/profiles $R/gpurun_out/r06_s26
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
python $R/bench.py --steps 20 --warmup 5 > $R/gpurun_out/profiles/r06_bench_default_plain.json 2> $R/gpurun_out/r06_s26/plain.err
python $R/bench.py --steps 127 --warmup 5 --no-rmse > $R/gpurun_out/profiles/r06_bench_127_passes.json 2>> $R/gpurun_out/r06_s26/err.log
python $R/bench.py --steps 1023 --warmup 5 --no-rmse --no-cpu --no-secondary --repeats 3 > $R/gpurun_out/profiles/r06_bench_1023_passes.json 2>> $R/gpurun_out/r06_s26/err.log
for f in r06_bench_default_plain r06_bench_127_passes r06_bench_1023_passes; do python -c "
import json,sys; d=json.load(open('$R/gpurun_out/profiles/$f.json')); r=d['roofline']; print('$f', round(d['value'],1), [round(x,1) for x in d['repeats']['values']], r['kernel'], round(r['frac'],3), {k: (v['ms'], round(v['frac'],3)) for k,v in r['per_kernel'].items()})"; done
