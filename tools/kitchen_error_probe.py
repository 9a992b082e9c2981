#!/usr/bin/env python3
"""GPU-box probe: error of this build's KITCHEN render against the reference's converged kitchen-reference.exr (tests/golden/
ref_kitchen_reference.npz) at the reference's own film size (700x400), over a ladder of sample counts, two seeds each — the data behind
bench.py's time_to_rmse block.  Writes gpurun_out/kitchen_error.json (+ the last image)."""
import json, os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))
import ppg_host  # noqa: E402

sys.path.insert(0, ROOT)
from bench import scene_props  # noqa: E402


def main():
    path = os.path.join(ROOT, "scratch", "kitchen-improved.ppgs")
    scene = ppg_host.load_scene_file(path)
    scene.camera = ppg_host.resize_camera(scene.camera, 700, 400)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_kitchen_reference.npz"))
    ref = fx["rgb"].astype(np.float64)
    props = scene_props(path, dict(budgetType="spp"))
    rows = []
    ladder = [int(a) for a in sys.argv[1:]] or [63, 127, 255, 511, 1023, 2400]
    ppg_host.GuidedPathTracer(engine=ppg_host.Engine.hip(**dict(props, budget=7))).render(scene)
    for spp in ladder:
        imgs, secs = [], []
        for seed in (1234, 98765):
            e = ppg_host.Engine.hip(**dict(props, budget=spp, seed=seed))
            t = time.time()
            imgs.append(ppg_host.GuidedPathTracer(engine=e).render(scene).astype(np.float64))
            secs.append(time.time() - t)
        row = dict(spp=spp, seconds=secs)
        for k, im in enumerate(imgs):
            d = im - ref
            row["rmse_%d" % k] = float(np.sqrt((d * d).mean()))
            row["mape_%d" % k] = float((np.abs(d) / (ref + 0.01)).mean())
            row["mean_rgb_%d" % k] = im.mean((0, 1)).tolist()
            row["max_%d" % k] = float(im.max())
        d = imgs[0] - imgs[1]
        row["split_half_rmse"] = float(np.sqrt((d * d).mean()) / np.sqrt(2))
        row["split_half_mape"] = float((np.abs(d) / (0.5 * (imgs[0] + imgs[1]) + 0.01)).mean() / np.sqrt(2))
        avg = 0.5 * (imgs[0] + imgs[1])
        d = avg - ref
        row["avg_rmse"] = float(np.sqrt((d * d).mean()))
        row["avg_mape"] = float((np.abs(d) / (ref + 0.01)).mean())
        rows.append(row)
        print(json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(targets={k: float(fx[k]) for k in fx.files if k.endswith(("rmse", "mape"))}, ref_mean_rgb=fx["mean_rgb"].tolist(), rows=rows),
              open(os.path.join(ROOT, "gpurun_out", "kitchen_error.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "kitchen_700x400.npz"), a=imgs[0].astype(np.float16), b=imgs[1].astype(np.float16))


if __name__ == "__main__":
    main()
