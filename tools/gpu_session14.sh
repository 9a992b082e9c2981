#!/bin/bash
set -x
mkdir -p gpurun_out/s14
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
timeout 300 python -X faulthandler bench.py --force-dist --steps 20 --warmup 5 $Q > gpurun_out/s14/bench_force_dist.json 2> gpurun_out/s14/bench_force_dist.log; echo "rc=$?" >> gpurun_out/s14/bench_force_dist.log
tail -25 gpurun_out/s14/bench_force_dist.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s14/ktrace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 0 $Q > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/s14/ktrace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = []
for r in rows:
    n = r['Kernel_Name']
    if n.startswith('void k_tail') or n.startswith('void k_commit'):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n.split('(')[0][5:], r.get('Queue_Id'), r.get('Stream_Id')))
ev.sort()
t0 = ev[0][0]
with open('gpurun_out/s14/tail_commit_timeline.txt', 'w') as o:
    for s, e, n, q, st in ev:
        o.write('%10.3f ms .. %10.3f ms  (%8.3f ms)  %-22s queue %s stream %s\n' % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, n, q, st))
print(open('gpurun_out/s14/tail_commit_timeline.txt').read())
PY
rm -rf gpurun_out/s14/ktrace
