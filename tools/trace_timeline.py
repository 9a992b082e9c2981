#!/usr/bin/env python3
"""Compact timeline of a rocprofv3 --kernel-trace CSV: one line per kernel launch of the LAST render in the trace (start, duration, gap to
the previous kernel's end, stream, short name, grid), then per-kernel totals and the idle time between kernels.
usage: trace_timeline.py p_kernel_trace.csv [--all] [--from-ms X]"""
import csv, sys, collections, re

def short(n):
    n = n.replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+)(<[^(]*>)?", n)
    s = m.group(1).split("::")[-1] + (m.group(2) or "") if m else n[:40]
    return s[:60]

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", r.get("Queue_Id")), int(r["Grid_Size_X"])) for r in rows]
# the last render = after the last gap > 200 ms?  simpler: find the last k_generate preceded by a k_build_grid with S-tree of one node: use the largest gaps
if "--all" not in sys.argv:
    # renders are separated by host-side scene set-up (BVH build): take everything after the last gap of > 50 ms
    cut = 0
    for i in range(1, len(ks)):
        if ks[i][0] - ks[i - 1][1] > 50e6:
            cut = i
    ks = ks[cut:]
t0 = ks[0][0]
tot = collections.defaultdict(float); cnt = collections.Counter()
prev_end = t0; busy_end = t0; idle = 0.0
for s, e, n, st, g in ks:
    gap = (s - busy_end) / 1e6
    if gap > 0:
        idle += gap
    print("%9.3f %8.3f gap %7.3f  s%s %-58s g=%d" % ((s - t0) / 1e6, (e - s) / 1e6, gap, st, n, g))
    busy_end = max(busy_end, e)
    tot[n] += (e - s) / 1e6; cnt[n] += 1
print("# total span %.3f ms, idle (no kernel running) %.3f ms" % ((busy_end - t0) / 1e6, idle))
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("# %-60s %9.3f ms %5d launches" % (n, v, cnt[n]))
