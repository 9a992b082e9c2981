"""Render a flat scene file (ppg_host.save_scene / `ppg_render --ppgs`) on the GPU: parity against the oracle at a small size, then
timed renders at the sizes given; images and a JSON summary go to gpurun_out/.  Used for the reference's bundled SPACESHIP scene
(converted in the development container, where the reference tree is mounted; the flat file lives in scratch/, untracked):

    python -m ppg_host /root/reference/scenes/spaceship/spaceship.xml --lenient --data-dir /root/reference/mitsuba/data --ppgs scratch/spaceship.ppgs
    gpurun -- python tools/scene_run.py scratch/spaceship.ppgs --sizes 640x360,1920x1080 --spp 1023

Sizes must keep the file's aspect ratio (sample_to_camera is reused).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--sizes", default="640x360")
    ap.add_argument("--spp", type=float, default=1023)
    ap.add_argument("--parity", default="320x180", help="size of the GPU = oracle check (empty: skip)")
    ap.add_argument("--parity-spp", type=float, default=31)  # >= 5 iterations: with 1 spp/pass the first iteration has no variance estimate and
    # inverse-variance combination over the last 4 iterations would make a 15-spp image NaN everywhere (in the reference too)
    ap.add_argument("--cpu-spp", type=float, default=31, help="oracle timing at the first size (0: skip)")
    ap.add_argument("--constant-env", help="R,G,B: add a constant environment emitter (STAND-IN lighting for a scene whose own emitter was skipped)")
    ap.add_argument("-P", dest="props", action="append", default=[])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out"))
    a = ap.parse_args()
    import torch  # noqa: F401  (HIP runtime first)
    import ppg_host
    from conftest import ORACLE_SO, make_oracle
    desc = ppg_host.load_scene_file(a.scene)
    if a.constant_env:
        desc.environment = tuple(float(v) for v in a.constant_env.split(","))
    props = {}
    pf = a.scene + ".props"
    if os.path.exists(pf):
        for line in open(pf):
            if "=" in line:
                k, v = line.strip().split("=", 1)
                props[k] = v
    for kv in a.props:
        k, v = kv.split("=", 1)
        props[k] = v
    for k, v in list(props.items()):
        if isinstance(v, str):
            try:
                props[k] = int(v)
            except ValueError:
                try:
                    props[k] = float(v)
                except ValueError:
                    props[k] = {"true": 1, "false": 0}.get(v, v)
    os.makedirs(a.out, exist_ok=True)
    name = os.path.splitext(os.path.basename(a.scene))[0]
    summary = dict(scene=name, stand_in_environment=a.constant_env, triangles=desc.n_triangles, spheres=len(desc.spheres), materials=len(desc.materials), props=props, runs=[])

    def sized(wh):
        w, h = (int(v) for v in wh.split("x"))
        desc.camera = ppg_host.resize_camera(desc.camera, w, h)  # keeps the horizontal field of view
        return w, h
    if a.parity:
        w, h = sized(a.parity)
        p = dict(props, budgetType="spp", budget=a.parity_spp)
        g = ppg_host.GuidedPathTracer(**p)
        ig = g.render(desc)
        o = ppg_host.GuidedPathTracer(engine=make_oracle(C.CDLL(ORACLE_SO), threads=os.cpu_count() or 8, **p))
        io = o.render(desc)
        same = bool(np.array_equal(ig, io, equal_nan=True))
        finite = float(np.isfinite(ig).mean())
        summary["parity"] = dict(size=a.parity, spp=a.parity_spp, bit_exact=same, finite_fraction=finite,
                                 max_abs_diff=float(np.nanmax(np.abs(ig - io))) if finite > 0 else None)
        print("parity", summary["parity"], flush=True)
    for k, wh in enumerate(a.sizes.split(",")):
        w, h = sized(wh)
        p = dict(props, budgetType="spp", budget=a.spp)
        g = ppg_host.GuidedPathTracer(**p)
        t0 = time.time()
        img = g.render(desc)
        dt = time.time() - t0
        samples = sum(st["samples"] for it in g.iterations for st in it.get("stats", [])) or w * h * a.spp
        rays = sum(st["rays"] for it in g.iterations for st in it.get("stats", []))
        run = dict(size=wh, spp=a.spp, seconds=dt, msamples_per_s=samples / dt / 1e6, mrays_per_s=rays / dt / 1e6, iterations=len(g.iterations),
                   mean_rgb=[float(v) for v in np.nanmean(img.reshape(-1, 3), 0)])
        summary["runs"].append(run)
        print("gpu", run, flush=True)
        np.save(os.path.join(a.out, "%s_%s_%dspp.npy" % (name, wh, int(a.spp))), img.astype(np.float16) if w * h > 1000000 else img)
        if k == 0 and a.cpu_spp > 0:
            p = dict(props, budgetType="spp", budget=a.cpu_spp)
            o = ppg_host.GuidedPathTracer(engine=make_oracle(C.CDLL(ORACLE_SO), threads=os.cpu_count() or 8, **p))
            t0 = time.time()
            o.render(desc)
            dt = time.time() - t0
            summary["cpu"] = dict(size=wh, spp=a.cpu_spp, seconds=dt, msamples_per_s=w * h * a.cpu_spp / dt / 1e6, threads=os.cpu_count())
            print("cpu", summary["cpu"], flush=True)
    json.dump(summary, open(os.path.join(a.out, "%s_run.json" % name), "w"), indent=1)


if __name__ == "__main__":
    main()
