#!/bin/bash
# GPU box, round 6, session 3: the whole GPU suite with the stragglers; a kernel timeline of the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_s3
mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/tests_all.log 2>&1
tail -5 $OUT/tests_all.log
cd /tmp; export TMPDIR=/tmp; ulimit -c 0
B="timeout 600 python $R/bench.py --warmup 5 --no-cpu --no-rmse --no-secondary --no-roofline --no-single-call"
$B --steps 20 > $OUT/b20.json 2> $OUT/b20.err
PPG_SPLIT_DEPTH=0 $B --steps 20 > $OUT/b20_nosplit.json 2>> $OUT/b20.err
for f in b20 b20_nosplit; do python -c "
import json; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['repeats']['values'])"; done
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $B --steps 20 --repeats 2 > $OUT/trace.json 2>> $OUT/b20.err
F=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_timeline.py $F > $OUT/timeline20.txt 2>&1
tail -30 $OUT/timeline20.txt
rm -rf $OUT/trace
