#!/usr/bin/env python3
"""GPU box: per-bounce latency of the persistent-thread tail in the SPARSE regime.  KITCHEN at a tiny film (a handful of paths per batch),
every batch goes to k_tail without a wavefront bounce (PPG_BULK_BOUNCES=0); k_tail's launch time / longest path = microseconds per bounce
of one lane when nothing else runs.  Run with PPG_DEBUG_BATCH=1 to get the longest path per iteration on stderr."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd")); sys.path.insert(0, ROOT)
import ppg_host
from bench import KITCHEN_FILE, scene_props

w, h, passes = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
scene = ppg_host.load_scene_file(KITCHEN_FILE)
scene.camera = ppg_host.resize_camera(scene.camera, w, h)
props = scene_props(KITCHEN_FILE, dict(budgetType="spp", seed=1234))
e = ppg_host.Engine.hip(budget=float(passes), **props)
e.set_scene(scene); e.enable_kernel_timing(True)
g = ppg_host.GuidedPathTracer(engine=e)
t0 = time.perf_counter(); g.render(); dt = time.perf_counter() - t0
kt = {k["name"]: k for k in e.kernel_times()}
rays = sum(s["rays"] for it in g.iterations for s in it["stats"])
out = {"film": [w, h], "passes": passes, "seconds": dt, "rays": rays, "iterations": [(it["passes"], sum(s["rays"] for s in it["stats"])) for it in g.iterations],
       "k_tail": kt.get("k_tail"), "us_per_ray_in_tail": 1e3 * kt["k_tail"]["ms"] / max(1, rays) if "k_tail" in kt else None}
print(json.dumps(out))
