#!/usr/bin/env python3
"""GPU box, ONE GPU: what does ONE rank of an N-rank render of bench.py's KITCHEN workload compute, and how long does it take?

The driver's 8-GPU run is the measurement of scaling; this probe is what a single MI355X can say about it beforehand.  Rank 0 of world
N = 1, 2, 4, 8 renders its share — its 32x32 tiles for the training passes, its share of the final iteration's groups — through the
sharded control flow of the library (ppg_set_shard, the round hook, the final groups' slots), ALONE: a stand-in reducer replaces every
exchange by what it would deliver statistically,
  * SD-tree statistics: this rank's integer sums times N (tiles are dealt round-robin, so every region of the scene is sampled by every
    rank: N times its own statistics is the all-reduced tree up to noise — same refinement, same D-tree sizes, same path lengths);
  * optimiser records: this rank applies its own (1 / N of every D-tree's records instead of all records of 1 / N of the D-trees: the same
    number of records through the same kernels);
  * images, final groups: nothing (no other rank's pixels; the variance estimate is not used by `inversevar` renders of fixed length).
So T(N) is the COMPUTE time of one rank incl. the tails it cannot shed and the host-side cost of the sharded control flow; the wire time of
the exchanges (~1.4 % of the single-GPU render at N = 8) is not in it.  T(1) / T(N) is the strong-scaling factor a
perfect interconnect would give.

    python tools/shard_scaling_probe.py [passes ...]      (default 20 127 1023)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd")); sys.path.insert(0, ROOT)
import torch
import ppg_host
from ppg_host.distributed import _view
from bench import KITCHEN_FILE, scene_props


class AloneReducer:
    """The exchanges of ppg_host.distributed.TorchReducer as seen by one rank whose peers are not there (see the module docstring)."""
    status = 0

    def __init__(self, world):
        self.world, self.device = world, torch.device("cuda", 0)

    def begin_render(self):
        self.status = 0

    def stop_decision(self, local_stop):
        return 1 if local_stop else 0

    def broadcast(self, v):
        return v

    def reduce_images(self, e):
        e.image_buffers()  # (the accessor's stream synchronisation is part of what a rank pays)

    def reduce_final_partials(self, e, ptr, count):
        e.final_partials_commit()

    def reduce_sdtree(self, e):
        for ptr, n in e.stat_buffers():
            if n:
                _view(torch, ptr, n, "<i8", self.device).mul_(self.world)
        torch.cuda.synchronize()

    def reduce_adam(self, e):
        pass  # the library applies this rank's own records

    def reduce_film(self, e, inverse_variance=False):
        pass


def main():
    passes_list = [int(a) for a in sys.argv[1:]] or [20, 127, 1023]
    scene = ppg_host.load_scene_file(KITCHEN_FILE)
    props = scene_props(KITCHEN_FILE, dict(budgetType="spp", seed=1234))
    W, H = scene.camera["width"], scene.camera["height"]
    out = {"workload": "kitchen-improved-720p", "what": "compute time of rank 0 of N, alone on one MI355X (tools/shard_scaling_probe.py)", "runs": []}

    def render(world, passes):
        e = ppg_host.Engine.hip(budget=float(passes), **props)
        e.set_scene(scene)
        red = None
        if world > 1:
            e.set_shard(0, world, 32)
            red = AloneReducer(world)
        g = ppg_host.GuidedPathTracer(engine=e, reducer=red)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.render()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        samples = sum(s["samples"] for it in g.iterations for s in it["stats"])
        rays = sum(s["rays"] for it in g.iterations for s in it["stats"])
        leaves = g.iterations[-1]["tree"]["n_leaves"]
        e.close()
        return dt, samples, rays, leaves

    render(1, 5)
    for passes in passes_list:
        base = None
        for world in (1, 2, 4, 8):
            render(world, min(passes, 20))
            runs = [render(world, passes) for _ in range(3 if passes <= 127 else 2)]
            dt = min(r[0] for r in runs)
            _, samples, rays, leaves = runs[-1]
            if world == 1:
                base = dt
            rec = dict(passes=passes, world=world, ms=round(dt * 1e3, 2), share_of_samples=round(samples / (W * H * passes), 4), rays_per_sample=round(rays / samples, 3),
                       leaves=leaves, compute_scaling=round(base / dt, 2))
            out["runs"].append(rec)
            print(json.dumps(rec), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "profiles", "r06_shard_scaling_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
