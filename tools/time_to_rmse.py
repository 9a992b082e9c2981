#!/usr/bin/env python3
"""time-to-target-RMSE (BASELINE.json metric, second half) on the GPU box.

1. converged reference: the HIP path at `--ref-spp` (different seed),
2. CPU baseline: the oracle (host cores) at `--cpu-spp`; its RMSE against the reference is the target,
3. GPU: render()s with budgets 1x, 1.5x, 2x ... of cpu-spp/4 upwards (own seed) until RMSE <= target; the wall time of
   the first budget that reaches it is the time-to-target.
RMSE is taken over the weight-normalised RGB film, per-pixel values clamped to 10 to keep single fireflies from
deciding the outcome (the reference's variance estimate clamps likewise, guided_path.cpp:1310).
Writes gpurun_out/profiles/r01_time_to_rmse.json.
"""
import argparse, ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))
import numpy as np
import torch  # noqa: F401
import ppg_host

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="cbox", choices=["cbox", "room"])
ap.add_argument("--scene-file", help="flat scene file (e.g. scratch/spaceship.ppgs) with its .props instead of a procedural scene")
ap.add_argument("--width", type=int, default=1280); ap.add_argument("--height", type=int, default=720)
ap.add_argument("--ref-spp", type=int, default=8188); ap.add_argument("--cpu-spp", type=int, default=60)
args = ap.parse_args()

props = dict(budgetType="spp", maxDepth=10, rrDepth=10, strictNormals=1)
if args.scene_file:
    scene = ppg_host.load_scene_file(args.scene_file)
    args.width, args.height = scene.camera["width"], scene.camera["height"]
    args.scene = os.path.splitext(os.path.basename(args.scene_file))[0]
    props = dict(budgetType="spp")
    for line in open(args.scene_file + ".props"):
        if "=" in line:
            k, v = line.strip().split("=", 1)
            if k not in ("budget", "budgetType"):
                props[k] = int(v) if v.lstrip("-").isdigit() else (float(v) if v.replace(".", "", 1).replace("-", "", 1).isdigit() else v)
elif args.scene == "cbox":
    scene = ppg_host.cbox_scene(args.width, args.height)
else:
    scene = ppg_host.room_scene(args.width, args.height, n_boxes=1820, tess=8)
    props = dict(budgetType="spp", sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic",
                 directionalFilter="box", sTreeThreshold=4000, sppPerPass=1, maxDepth=-1, rrDepth=5)


def rmse(a, b):
    return float(np.sqrt(np.mean((np.minimum(a, 10.0) - np.minimum(b, 10.0)) ** 2)))


def gpu(budget, seed):
    e = ppg_host.Engine.hip(budget=float(budget), seed=seed, **props)
    e.set_scene(scene)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.render()
    img = e.read_film()
    return img, time.perf_counter() - t0


ref, t_ref = gpu(args.ref_spp, 777)
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libppg_oracle.so"))
cores = os.cpu_count() or 1
o = ppg_host.Engine(lib, "ppgo_", budget=float(args.cpu_spp), seed=1, **props)
lib.ppgo_set_modes(o.ctx, 0, 0, cores)
o.set_scene(scene)
t0 = time.perf_counter(); o.render(); t_cpu = time.perf_counter() - t0
target = rmse(o.read_film(), ref)
gpu(4, 5)  # warm-up
trials = []
budget = max(4, args.cpu_spp // 4)
while True:
    img, t = gpu(budget, 99)
    r = rmse(img, ref)
    trials.append(dict(spp=budget, seconds=t, rmse=r))
    if r <= target or budget > 8 * args.cpu_spp:
        break
    budget = int(budget * 1.5) + 1
res = dict(scene=args.scene, resolution=[args.width, args.height], reference_spp=args.ref_spp, reference_seconds=t_ref,
           cpu=dict(kind="port (oracle)", cores=cores, spp=args.cpu_spp, seconds=t_cpu, rmse=target),
           gpu_trials=trials, gpu_time_to_target_s=trials[-1]["seconds"], speedup_at_equal_rmse=t_cpu / trials[-1]["seconds"])
os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "profiles", "r01_time_to_rmse_%s.json" % args.scene), "w"), indent=1)
print(json.dumps(res))
