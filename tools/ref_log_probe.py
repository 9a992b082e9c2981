#!/usr/bin/env python3
"""GPU box: per-iteration SD-tree statistics of the reference's own scenes at the reference's own film sizes — what its render logs print
(GP:1176-1186: depth, mean radiance, node count, statistical weight, each as [min, avg, max]; tests/golden/ref_logs.json holds them) —
from this build, for several seeds: the spread that tolerances of tests/test_real_scenes.py::test_tree_statistics_follow_the_reference_logs
are stated from.

    python tools/ref_log_probe.py [passes]   →  gpurun_out/ref_log_probe.json"""
import copy, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd")); sys.path.insert(0, ROOT)
import torch  # noqa: F401
import ppg_host
from bench import scene_props

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 31
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_logs.json")))["scenes"]
out = {}
for name, path, log in (("kitchen-improved", "scratch/kitchen-improved.ppgs", "kitchen-improved"), ("spaceship", "scratch/spaceship.ppgs", "spaceship")):
    full = os.path.join(ROOT, path)
    if not os.path.exists(full):
        continue
    scene = copy.copy(ppg_host.load_scene_file(full))
    w, h = ref[log]["width"], ref[log]["height"]
    scene.camera = ppg_host.resize_camera(scene.camera, w, h)
    runs = []
    for seed in (1234, 99, 7, 20260927):
        props = scene_props(full, dict(budgetType="spp", seed=seed))
        e = ppg_host.Engine.hip(budget=float(passes * int(props.get("sppPerPass", 4))), **props)
        g = ppg_host.GuidedPathTracer(engine=e)
        g.render(scene)
        runs.append([dict(iter=it["iter"], passes=it["passes"], var=it["stats"][-1]["variance"], vertices=sum(s["vertices_committed"] for s in it["stats"]),
                          samples=sum(s["samples"] for s in it["stats"]), rays=sum(s["rays"] for s in it["stats"]), **it["tree"]) for it in g.iterations])
        e.close()
    out[name] = {"film": [w, h], "runs": runs, "reference": ref[log]["iterations"][:len(runs[0])], "spp_per_pass": int(props.get("sppPerPass", 4))}
    for k, it in enumerate(runs[0]):
        r = ref[log]["iterations"][k] if k < len(ref[log]["iterations"]) else None
        print(name, "iter", k, "passes", it["passes"], "| leaves", [run[k]["n_leaves"] for run in runs], "| avg sw", [round(run[k]["avg_stat_weight"], 1) for run in runs],
              "| max sw", [round(run[k]["max_stat_weight"]) for run in runs], "| avg depth", [round(run[k]["avg_depth"], 3) for run in runs], "| avg nodes", [round(run[k]["avg_nodes"], 2) for run in runs],
              "| avg rad", [round(run[k]["avg_mean_radiance"], 5) for run in runs], "| var", [round(run[k]["var"], 4) for run in runs])
        if r:
            print("      reference: sw", r["stat_weight"], "depth", r["depth"], "nodes", r["node_count"], "rad", r["mean_radiance"], "var", r["var"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ref_log_probe.json"), "w"), indent=1)
