#!/bin/bash
set -x
mkdir -p gpurun_out/s21
export TMPDIR=/tmp
Q="--no-cpu --no-rmse --no-secondary --no-roofline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/s21/ktrace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 $Q > $GRAFT_REPO_ROOT/gpurun_out/s21/bench.json 2>/dev/null
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/s21/ktrace/**/*kernel_trace.csv', recursive=True)[0]
ev = []
for r in csv.DictReader(open(f)):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]))
for g in glob.glob('gpurun_out/s21/ktrace/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(g)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')))
ev.sort()
# the timed render = the last k_generate-started sequence: find the last big gap (> 100 ms?) no: print the last 900 events' gaps
t_end = ev[-1][1]
# locate start of the last render: the last launch of k_dtree_reset... simpler: take events in the last 400 ms
sel = [e for e in ev if e[0] > t_end - 400_000_000]
out = open('gpurun_out/s21/gaps.txt', 'w')
t0 = sel[0][0]
busy_until = sel[0][0]
for s, e, n in sel:
    if s - busy_until > 300_000:
        out.write('GAP %8.3f ms before %-40s at %9.3f ms\n' % ((s - busy_until) / 1e6, n, (s - t0) / 1e6))
    busy_until = max(busy_until, e)
out.write('span %.3f ms, events %d\n' % ((sel[-1][1] - t0) / 1e6, len(sel)))
# per-kernel totals in the window
import collections
tot = collections.Counter()
for s, e, n in sel: tot[n] += e - s
for n, v in tot.most_common(14): out.write('%-42s %9.3f ms\n' % (n, v / 1e6))
out.close()
print(open('gpurun_out/s21/gaps.txt').read())
PY
rm -rf gpurun_out/s21/ktrace
