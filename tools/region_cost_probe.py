"""GPU box: what rounds by image region (include/ppg.h ppg_set_adam_regions) cost on KITCHEN at 1280x720 — complete renders of 20 and 127 passes
with 0 (default), 8 and 16 groups per pass in the iterations of up to 16 passes.  python tools/region_cost_probe.py"""
import sys, os, time
sys.path.insert(0, "/root/repo/practical-path-guiding_amd"); sys.path.insert(0, "/root/repo")
import torch, ppg_host
from bench import KITCHEN_FILE, scene_props
scene = ppg_host.load_scene_file(KITCHEN_FILE)
props = scene_props(KITCHEN_FILE, dict(budgetType="spp", seed=1234))
def run(passes, regions):
    e = ppg_host.Engine.hip(budget=float(passes), **props); e.set_scene(scene); e.set_adam_regions(regions)
    g = ppg_host.GuidedPathTracer(engine=e); torch.cuda.synchronize(); t = time.perf_counter(); g.render(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    e.close(); return dt
run(5, 0)
for passes in (20, 127):
    for regions in (0, 8, 16):
        run(passes, regions)
        print("passes", passes, "regions", regions, "ms %.1f" % (min(run(passes, regions) for _ in range(2)) * 1e3), flush=True)
