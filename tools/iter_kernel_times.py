"""Per-iteration kernel time breakdown of the default bench workload (HIP-event timing of every launch; the sync per
launch makes the total slower than bench.py, the split is what matters)."""
import sys
sys.path.insert(0, '/root/repo/practical-path-guiding_amd')
import torch, ppg_host  # noqa
props = dict(budgetType="spp", sppPerPass=4, maxDepth=10, rrDepth=10, strictNormals=1, seed=1234, budget=4.0 * 255)
scene = ppg_host.cbox_scene(1280, 720)
e = ppg_host.Engine.hip(**props); e.set_scene(scene)
e.enable_kernel_timing(True)
e.begin_render()
prev = {}
passes = [1, 2, 4, 8, 16, 32, 64, 128]
for it, p in enumerate(passes):
    e.begin_iteration(it == len(passes) - 1)
    e.render_passes_nostat(p); e.finish_passes(); e.build_sdtree(); e.end_iteration()
    cur = {k["name"]: (k["ms"], k["launches"], k["units"]) for k in e.kernel_times()}
    row = []
    for n, (ms, l, u) in cur.items():
        pm, pl, pu = prev.get(n, (0, 0, 0))
        if ms - pm > 0.005:
            row.append("%s %.2fms/%d (%.1f M units)" % (n, ms - pm, l - pl, (u - pu) / 1e6))
    print("iter %d (%d passes):" % (it, p), "; ".join(row), flush=True)
    prev = cur
e.end_render()
