#!/usr/bin/env python3
"""GPU box: invariants of the stragglers' machinery at the bench's full size (KITCHEN scene-improved, 1280x720, 20 passes) — no oracle needed:
  * scheduling-only splits must not show in any result: the final iteration in two launches with stragglers handed over at depth 8
    (PPG_FINAL_HALVES=1 PPG_SPLIT_DEPTH=8) gives the picture, the SD-tree and the counters of the default schedule bit for bit;
  * run-to-run determinism with a THIRD of all paths as stragglers (the tests' switch: depth 8), and that render differs from the default one
    only through the learned fractions (same first iteration — rendered before any round —, same sample counts).
    python tools/full_size_invariance_probe.py"""
import ctypes as C, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd")); sys.path.insert(0, ROOT)

WORKER = r'''
import ctypes as C, os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1] + "/practical-path-guiding_amd"); sys.path.insert(0, sys.argv[1])
import torch, ppg_host
from bench import KITCHEN_FILE, scene_props
scene = ppg_host.load_scene_file(KITCHEN_FILE)
props = scene_props(KITCHEN_FILE, dict(budgetType="spp", seed=1234))
e = ppg_host.Engine.hip(budget=20.0, **props)
if int(sys.argv[3]):
    e._call("debug_set_defer_depth", C.c_int32(int(sys.argv[3])))
g = ppg_host.GuidedPathTracer(engine=e)
img = g.render(scene)
t = e.read_sdtree()
np.savez(sys.argv[2], img=img, theta=t["theta"], children=t["children"], sums=t["sampling"]["node_sums"],
         stats=np.array([[s["rays"], s["path_length_sum"], s["vertices_committed"], s["samples"]] for it in g.iterations for s in it["stats"]], np.uint64))
'''


def run(tag, env, depth=0):
    out = "/tmp/inv_%s.npz" % tag
    open("/tmp/inv_worker.py", "w").write(WORKER)
    subprocess.run([sys.executable, "/tmp/inv_worker.py", ROOT, out, str(depth)], env=dict(os.environ, **env), check=True, stderr=subprocess.DEVNULL)
    return np.load(out)


def same(a, b):
    return all(np.array_equal(a[k], b[k], equal_nan=True) for k in a.files)


base = run("base", {})
halves = run("halves", dict(PPG_FINAL_HALVES="1", PPG_SPLIT_DEPTH="8"))
nos = run("nosplit", dict(PPG_SPLIT_DEPTH="0", PPG_NO_OVERLAP="1"))
d8a, d8b = run("d8a", {}, 8), run("d8b", {}, 8)
res = {"final_iteration_in_two_launches_with_stragglers_at_depth_8_equals_default": same(base, halves),
       "one_stream_no_scheduling_split_equals_default": same(base, nos),
       "a_third_of_all_paths_as_stragglers_is_deterministic": same(d8a, d8b),
       "and_differs_from_the_default_from_the_first_round_on_only": bool(not np.array_equal(d8a["theta"], base["theta"]) and np.array_equal(d8a["stats"][:1], base["stats"][:1])
                                                                        and np.array_equal(d8a["stats"][:, 3], base["stats"][:, 3]))}
# (a 20-spp picture of this preset is NaN — inverse-variance combination over the one-sample first iteration, in the reference as well; the
# comparisons above are bit comparisons, NaN pattern included)
print(json.dumps(res))
assert all(res.values()), res
