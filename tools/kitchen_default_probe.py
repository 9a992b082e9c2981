#!/usr/bin/env python3
"""GPU box: KITCHEN with the DEFAULT settings of scenes/kitchen/kitchen.xml (4 spp per pass, nearest filters, no learned fraction, automatic
sample combination, 2400 spp) at the reference's 700x400, three seeds — against the reference's converged kitchen-reference.exr (MAPE / RMSE over
the pixels outside the missing meshes' footprint, next to the reference's own kitchen.exr figures) and against kitchen.exr itself in 50x50-pixel
block means (tests/golden/ref_kitchen_reference.npz).  The data behind tests/test_real_scenes.py::test_kitchen_default_configuration_picture."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))
import ppg_host


def main():
    scene = ppg_host.load_scene_file(os.path.join(ROOT, "scratch", "kitchen-improved.ppgs"))  # (the two XMLs differ in the integrator block only)
    scene.camera = ppg_host.resize_camera(scene.camera, 700, 400)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_kitchen_reference.npz"))
    ref = fx["rgb"].astype(np.float64)
    blk = int(fx["mask_block"])
    keep = ~np.kron(fx["mask_blocks"], np.ones((blk, blk), np.uint8)).astype(bool)
    keep50 = keep.reshape(8, 50, 14, 50).all((1, 3))
    rows = []
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
    ppg_host.GuidedPathTracer(engine=ppg_host.Engine.hip(budgetType="spp", strictNormals=1, budget=28)).render(scene)
    for seed in (1234, 98765, 4321):
        e = ppg_host.Engine.hip(budgetType="spp", strictNormals=1, budget=spp, seed=seed)  # kitchen.xml:4-18
        t = time.time()
        g = ppg_host.GuidedPathTracer(engine=e)
        im = g.render(scene).astype(np.float64)
        dt = time.time() - t
        d = (im - ref)[keep]
        b50 = im.reshape(8, 50, 14, 50, 3).mean((1, 3))
        rel = np.abs(b50.mean(-1) / fx["kitchen_blocks50"].astype(np.float64).mean(-1) - 1)
        rows.append(dict(seed=seed, seconds=dt, passes=[it["passes"] + it.get("final_passes", 0) for it in g.iterations], mape=float((np.abs(d) / (ref[keep] + 0.01)).mean()),
                         rmse=float(np.sqrt((d * d).mean())), block_rel_max_unmasked=float(rel[keep50].max()), block_rel_mean_unmasked=float(rel[keep50].mean()),
                         block_rel_max_all=float(rel.max()), mean_rgb=im.mean((0, 1)).tolist()))
        print(json.dumps(rows[-1]), flush=True)
    out = dict(reference=dict(mape=float(fx["kitchen_mape_unmasked"]), rmse=float(fx["kitchen_rmse_unmasked"])), unmasked_blocks=int(keep50.sum()), rows=rows)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "profiles", "r06_kitchen_default_picture.json"), "w"), indent=1)
    print(json.dumps(out["reference"]), "unmasked 50x50 blocks:", int(keep50.sum()))


if __name__ == "__main__":
    main()
