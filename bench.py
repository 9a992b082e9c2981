#!/usr/bin/env python3
"""bench.py — Msamples/s of the guided path tracer hot path on MI355X.

Workload (BASELINE.json configs[1]): procedural CBOX (= scenes/cbox/cbox.xml, 36 triangles) at 1280x720,
4 spp per pass, default SD-tree parameters (sTreeThreshold 12000, dTreeThreshold 0.01,
bsdfSamplingFraction 0.5, nearest filters, sampleCombination automatic), maxDepth 10 / rrDepth 10 /
strictNormals as in the scene file, budgetType = spp.

A "step" is one render pass = one BlockedRenderProcess of the reference: every pixel x sppPerPass paths
through Li, splatted into the SD-tree, accumulated into the film.  The timed region is a complete
GuidedPathTracer::render() of K passes (budget = K x 4 spp) following the reference's iteration schedule
1, 2, 4, ... (guided_path.cpp:1342-1426), so SD-tree refine / reset / build between iterations ARE inside
the timed region and scene upload / BVH build are not.  Warm-up = one throw-away
render of W passes.  value = pixels x spp x K / seconds, whole job over all GPUs.

Multi-GPU (`torchrun ... bench.py --gpus N`): the fixed image is sharded by 32x32 tiles over the ranks
(strong scaling); per iteration the building SD-tree statistics are all-reduced over RCCL
(ppg_host/distributed.py).

Adds to the JSON line: `roofline` for the dominant kernel (HIP-event durations measured in-process on
the kernel's own stream; algorithmic bytes per DESIGN.md) and, on rank 0 at N = 1, `cpu_baseline` = the
oracle restatement timed on the host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def algorithmic_bytes(work=None, rays=None):
    """ALGORITHMIC bytes per unit of each path kernel: bytes the kernel's algorithm touches per
    unit, cache-oblivious, from the data layout of ppg_kernels.h and the operation counts measured by the CPU
    restatement on the same workload (`work` = ppgo_work_counters, `rays` = rays traced in that run).
    Without counters the a-priori CBOX values of a 63-pass run are used."""
    if work and rays:
        lookups = work[1] / rays                       # S-tree lookups (= bounces sampled) per traced ray
        ds = work[2] / rays                            # D-tree levels descended while sampling, per ray
        dp = work[4] / rays                            # ... while evaluating the pdf, per ray
    else:
        lookups, ds, dp = 0.70, 1.31, 2.48
    state_rd = 4 + 5 * 16                              # queue entry, ray_d, thr, li, hit, misc
    state_wr = 5 * 16 * lookups                        # ray_o, ray_d, thr, li (per surviving path) + misc
    shade = (state_rd + state_wr + 48 + 16             # + hit triangle (3 float4) + material
             + lookups * (4 + 64)                      # grid cell + leaf header
             + (ds + dp) * 32                          # sampling-tree nodes
             + lookups * 64)                           # speculative vertex record (4 float4)
    return {"k_shade": shade,
            "k_trace": 32 + 16,                        # small scene: ray in, hit out (triangles are LDS resident)
            "k_commit": 16 + 64 + 5 * 8 + 16,          # per recorded vertex: misc, vertex, descent, 2 atomics
            "k_generate": 80, "k_film": 4 * 16 + 88,
            "detail": {"lookups_per_ray": lookups, "dtree_sample_levels_per_ray": ds, "dtree_pdf_levels_per_ray": dp}}


def measured_traffic():
    """HBM bytes per unit from profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes,
    calibrated as MI355X_MICROARCH.md prescribes; written by tools/collect_profiles.py on the GPU box)."""
    p = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        return json.load(open(p))
    except Exception:
        return None


def run(args):
    import numpy as np
    import ppg_host

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.gpus > 1 or args.force_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl")
        assert dist.get_world_size() == args.gpus, "launch with --nproc-per-node == --gpus"
        world = dist.get_world_size()
    import torch  # noqa: F811  (device sync + barrier plumbing only)

    spp = args.spp
    props = dict(budgetType="spp", sppPerPass=spp, maxDepth=10, rrDepth=10, strictNormals=1, seed=1234, device=local_rank)
    workload = "cbox-720p: procedural CBOX (36 tris), %dx%d, %d spp/pass, %d passes, default SD-tree params, maxDepth 10" % (
        args.width, args.height, spp, args.steps)
    if args.scene_file:
        # a converted scene (python -m ppg_host scene.xml --ppgs FILE, e.g. the reference's SPACESHIP): its own integrator settings
        # (FILE.props) and film size; extra bench line, not the headline configuration
        scene = ppg_host.load_scene_file(args.scene_file)
        if args.size_override:
            scene.camera = dict(scene.camera, width=args.width, height=args.height)
        if args.constant_env:
            scene.environment = tuple(float(v) for v in args.constant_env.split(","))
        args.width, args.height = scene.camera["width"], scene.camera["height"]
        if os.path.exists(args.scene_file + ".props"):
            for line in open(args.scene_file + ".props"):
                if "=" in line:
                    k, v = line.strip().split("=", 1)
                    if k not in ("budget", "budgetType"):
                        props[k] = int(v) if v.lstrip("-").isdigit() else (float(v) if v.replace(".", "", 1).replace("-", "", 1).isdigit() else v)
        spp = int(props.get("sppPerPass", spp))
        workload = "%s (%d triangles, %d spheres), %dx%d, %d spp/pass, %d passes, the scene file's integrator settings" % (
            os.path.basename(args.scene_file), scene.n_triangles, len(scene.spheres), args.width, args.height, spp, args.steps)
    elif args.scene == "torus":
        # BASELINE.json configs[4] "TORUS (SDS caustics), 1920x1080, sppPerPass=1, sTreeThreshold=4000": the paper's scene is not bundled with
        # the reference — the labelled procedural stand-in of SURVEY.md §8(d) S5 (diffuse torus in a glass cube, one small emitter)
        if (args.width, args.height) == (1280, 720):
            args.width, args.height = 1920, 1080
        spp = 1
        scene = ppg_host.torus_scene(args.width, args.height)
        props.update(sppPerPass=1, sTreeThreshold=4000, maxDepth=-1, rrDepth=5, strictNormals=0)
        workload = "torus-1080p (torus-class STAND-IN: diffuse torus in a glass cube, %d triangles), %dx%d, 1 spp/pass, %d passes, sTreeThreshold 4000" % (
            scene.n_triangles, args.width, args.height, args.steps)
    elif args.scene == "cbox":
        scene = ppg_host.cbox_scene(args.width, args.height)
    else:
        # BASELINE.json configs[2] "kitchen-class improved": the bundled KITCHEN lacks 6 meshes and cannot travel to the GPU
        # box, so this is the labelled procedural stand-in of SURVEY.md §8(d) S3 (Lambertian only) with the README's
        # "improved" preset; maxDepth -1 / rrDepth 5 as in kitchen-improved.xml.
        scene = ppg_host.room_scene(args.width, args.height, n_boxes=args.room_boxes, tess=8, glossy=args.glossy)
        props.update(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box",
                     sTreeThreshold=4000, maxDepth=-1, rrDepth=5, strictNormals=0)
        workload = "room-720p (kitchen-class STAND-IN, %d %s triangles), %dx%d, %d spp/pass, %d passes, improved preset" % (
            scene.n_triangles, "Lambertian / GGX(0.1) / plastic" if args.glossy else "Lambertian", args.width, args.height, spp, args.steps)

    def make(budget_passes, timing=False):
        e = ppg_host.Engine.hip(budget=float(budget_passes * spp), **props)
        e.set_scene(scene)
        if world > 1:
            e.set_shard(rank, world, 32)
        if timing:
            e.enable_kernel_timing(True)
        red = None
        if dist is not None:
            from ppg_host.distributed import TorchReducer
            red = TorchReducer(dist, torch.device("cuda", local_rank))
        return ppg_host.GuidedPathTracer(engine=e, reducer=red)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    if args.warmup > 0:
        make(args.warmup).render()
    gpt = make(args.steps)
    sync()
    t0 = time.perf_counter()
    gpt.render()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    samples = args.width * args.height * spp * args.steps
    rays = sum(s["rays"] for it in gpt.iterations for s in it["stats"])
    own_samples = sum(s["samples"] for it in gpt.iterations for s in it["stats"])
    var_last = gpt.iterations[-1]["stats"][-1]["variance"]
    out = {
        "metric": "Msamples/s", "value": samples / dt / 1e6, "unit": "Msamples/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload,
                   "iterations": [it["passes"] for it in gpt.iterations], "parallelism": "tiles%d" % args.gpus,
                   "rays_per_sample": rays / max(1, own_samples), "variance_last_iteration": var_last},
    }

    work = rays_cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu:
        # CPU baseline: the oracle (a port, not the reference binary) on the host cores, bounded sample:
        # the same scene/resolution/settings, the first `cpu_passes` passes of the same schedule.
        import ctypes
        cores = os.cpu_count() or 1
        lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libppg_oracle.so"))
        cp = args.cpu_passes
        o = ppg_host.Engine(lib, "ppgo_", budget=float(cp * spp), **{k: v for k, v in props.items() if k != "device"})
        lib.ppgo_set_modes(o.ctx, 0, 0, cores)
        og = ppg_host.GuidedPathTracer(engine=o)
        o.set_scene(scene)
        t1 = time.perf_counter()
        og.render()
        dtc = time.perf_counter() - t1
        wk = (ctypes.c_uint64 * 8)()
        lib.ppgo_work_counters(o.ctx, wk)
        work = list(wk)
        rays_cpu = sum(s["rays"] for it in og.iterations for s in it["stats"])
        out["cpu_baseline"] = {"value": args.width * args.height * spp * cp / dtc / 1e6, "unit": "Msamples/s", "cores": cores,
                               "kind": "port", "sample": "first %d passes (%d spp) of the same render(), oracle restatement, OpenMP over 32x32 blocks"
                               % (cp, cp * spp), "seconds": dtc}

    times = None
    if not args.no_roofline:
        # separate instrumented render: per-kernel HIP-event durations on the kernels' stream.  EVERY rank runs it (the render
        # contains collectives); rank 0 reports its own kernels.
        g2 = make(args.steps, timing=True)
        g2.render()
        times = g2.engine.kernel_times()
        sync()
    if rank == 0 and times:
        dom = max(times, key=lambda k: k["ms"])
        alg = algorithmic_bytes(work, rays_cpu)
        name = dom["name"].split("<")[0]
        bytes_per_unit = alg.get(name, alg["k_shade"])
        avg_units = dom["units"] / max(1, dom["launches"])
        avg_ms = dom["ms"] / max(1, dom["launches"])
        achieved = bytes_per_unit * avg_units / (avg_ms * 1e-3) / 1e9
        tr = measured_traffic()
        traffic = None
        if tr and name in tr.get("bytes_per_unit", {}):
            traffic = tr["bytes_per_unit"][name] * avg_units
        out["roofline"] = {"bound": "hbm", "kernel": dom["name"], "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_ms, "launches": dom["launches"],
                           "algorithmic_bytes_per_unit": bytes_per_unit, "avg_units_per_launch": avg_units, "unit_of_work": "traced ray",
                           "operation_counts": alg["detail"], "kernels_ms": {k["name"]: round(k["ms"], 3) for k in times},
                           "note": "tree/scene bytes are cache resident: `traffic` (PMC) is what actually reaches HBM"
                                   + ("; k_trace on a BVH scene: the algorithmic count holds only the ray read and the hit written (48 B), node and "
                                      "triangle reads are not counted" if name == "k_trace" and scene.n_triangles > 64 else "")}

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:  # the one JSON line is the last thing written
        sys.stdout.flush(); sys.stderr.flush()
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=255, help="render passes in the timed render() (budget = steps * spp)")
    ap.add_argument("--warmup", type=int, default=3, help="passes of the throw-away warm-up render")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--scene", choices=["cbox", "room", "torus"], default="cbox", help="room = kitchen-class procedural stand-in, improved preset; torus = torus-class stand-in (SDS caustics)")
    ap.add_argument("--scene-file", help="flat scene file (ppg_host.save_scene / `python -m ppg_host scene.xml --ppgs`) instead of a procedural scene")
    ap.add_argument("--size-override", action="store_true", help="with --scene-file: render at --width x --height instead of the file's film size")
    ap.add_argument("--constant-env", help="with --scene-file: R,G,B of a constant environment emitter (STAND-IN lighting)")
    ap.add_argument("--room-boxes", type=int, default=1820, help="boxes of the room scene (768 triangles each)")
    ap.add_argument("--glossy", action="store_true", help="room scene with the S3 material mix (GGX alpha 0.1 metal, plastic) instead of Lambertian only")
    ap.add_argument("--cpu-passes", type=int, default=7, help="passes timed on the CPU baseline (bounded sample)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the reducer even with one rank (plumbing check)")
    run(ap.parse_args())


if __name__ == "__main__":
    main()
