#!/usr/bin/env python3
"""Timing probe: W contexts on ONE GPU, each rendering a tile shard of the bench workload concurrently (one host thread and one pair of
streams per context, no reducer: every shard learns from its own samples only — a probe of how well the GPU overlaps the kernels of
independent renders, not a product path).  Prints total samples / wall time for W = 1, 2, 3, 4.
usage: overlap_probe.py [passes] [worlds...]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))
import bench, ppg_host, torch

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worlds = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
scene = ppg_host.load_scene_file(bench.KITCHEN_FILE)
props = bench.scene_props(bench.KITCHEN_FILE, dict(budgetType="spp", seed=1234, device=0))
spp = int(props.get("sppPerPass", 1))
W, H = scene.camera["width"], scene.camera["height"]

def make(rank, world, n):
    e = ppg_host.Engine.hip(budget=float(n * spp), **props)
    e.set_scene(scene)
    if world > 1:
        e.set_shard(rank, world, 32)
    return ppg_host.GuidedPathTracer(engine=e)

make(0, 1, 5).render()
for world in worlds:
    for rep in range(2):
        g = [make(r, world, passes) for r in range(world)]
        torch.cuda.synchronize()
        th = [threading.Thread(target=x.render) for x in g]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("world %d rep %d: %.1f ms, %.1f Msamples/s" % (world, rep, dt * 1e3, W * H * spp * passes / dt / 1e6), flush=True)
        del g
