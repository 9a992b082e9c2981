#!/usr/bin/env python3
"""bench.py — Msamples/s (+ time to target RMSE) of the guided path tracer hot path on MI355X.

Default workload = BASELINE.json configs[2], the configuration the metric is quoted on: the reference's bundled KITCHEN scene
(scenes/kitchen/kitchen-improved.xml: the README's "improved" preset — inverse-variance combination, KL-learned BSDF sampling
fraction, stochastic + box filters, sTreeThreshold 4000, 1 spp per pass, unbounded path depth) at 1280x720.  The scene is read from
scratch/kitchen-improved.ppgs, the flat conversion of the XML made in the development container (`python -m ppg_host
kitchen-improved.xml --lenient --data-dir <mitsuba>/data --size 1280x720 --ppgs ...`: 283 of its 289 meshes — six are missing from
the reference checkout —, its eleven bitmap textures, its sunsky emitter baked into a radiance map).  If that file is absent the
labelled procedural stand-in `room` (1.4 M Lambertian triangles, same preset) is rendered instead and `config.workload` says so.
`--scene cbox` is configs[1] (procedural CBOX, 4 spp per pass, default parameters), `--scene torus` the configs[4] stand-in.

A "step" is one render pass = one BlockedRenderProcess of the reference: every pixel x sppPerPass paths through Li, recorded into
the SD-tree, accumulated into the film.  The timed region is a complete GuidedPathTracer::render() of K passes following the
reference's iteration schedule 1, 2, 4, ... (guided_path.cpp:1342-1426): SD-tree refine / reset / build between iterations and the
rounds of the sampling-fraction optimiser ARE inside the timed region; scene upload and BVH build are not.
Warm-up = one throw-away render of W passes.  value = pixels x spp x K / seconds, whole job over all GPUs.

Multi-GPU (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`, or plain `python bench.py --gpus N`, which starts
its N ranks itself): the fixed image is sharded by 32x32 tiles over the ranks (strong scaling); per iteration the building SD-tree
statistics are all-reduced over RCCL, per round of the optimiser its records go to the owners of their D-trees, the final iteration's
groups of passes are dealt whole to the ranks or rendered by tiles (ppg_host/distributed.py, include/ppg.h).

Added to the JSON line (rank 0):
  roofline      the kernel with the largest accumulated time of an instrumented render of the same K passes: HIP-event durations on
                the kernels' own stream, algorithmic bytes per DESIGN.md §3 — for k_trace on a BVH scene 48 B per ray + 64 B per
                BVH4 node visited + 48 B per triangle tested, the visits counted by the kernel itself in that run
  cpu_baseline  the oracle restatement timed on the host cores on the first passes of the same render — the GPU's own schedule when
                --steps <= 20 — and `one_core`: one thread on a film shrunk to 1/8 x 1/8 (N = 1 only)
  reference_log the reference's OWN run of this configuration (its embedded render log: 1.34 Msamples/s on 16 CPU cores, the complete scene)
                and vs_reference_log = value / that — the honest denominator; the triangle ratio of the checkout's scene is stated
  single_call   the same render through ppg_render(), the ONE C-ABI call a host in any language makes (no Python between the phases)
  tuning_env    every PPG_* variable set in the environment (none = the defaults the library ships with)
  time_to_rmse  seconds until this build's KITCHEN picture (at the reference's 700x400) is as close to the reference's converged
                kitchen-reference.exr as the reference's own guided render kitchen-improved.exr is (2400 spp, 500.9 s on 16 CPU
                cores): MAPE and RMSE over the pixels not affected by the six missing meshes; the render that meets the target is
                repeated with three seeds and the spread reported (N = 1 only; --no-rmse skips it)
  secondary     cbox-720p (configs[1]) Msamples/s, for the record (N = 1 only; --no-secondary skips it)
"""
import argparse
import gc
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
KITCHEN_FILE = os.path.join(ROOT, "scratch", "kitchen-improved.ppgs")
IMPROVED = dict(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=4000, sppPerPass=1)
REF_KITCHEN_SECONDS = 500.9  # render time in the log embedded in kitchen-improved.exr (2400 spp, 700x400, 16 CPU cores)
# The reference's own run of this configuration (the log embedded in scenes/kitchen/kitchen-improved.exr): the honest
# denominator — the reference BINARY on the COMPLETE scene — next to cpu_baseline, which times this repository's CPU restatement.
REF_KITCHEN_LOG = {"msamples_per_s": 1.34, "seconds": 500.9, "samples": 672.00e6, "rays": 4.327e9, "cores": 16, "os": "Windows", "film": "700x400",
                   "spp": 2400, "primitives": 1414390, "source": "scenes/kitchen/kitchen-improved.exr, log attribute (BASELINE.md §1)"}
KITCHEN_REFERENCE = os.path.join(ROOT, "tests", "golden", "ref_kitchen_reference.npz")


def stored_operation_counts(tag):
    """S-tree lookups / D-tree levels per ray of the same workload from the newest committed bench line that measured them with the CPU
    restatement (profiles/rNN_bench_default.json) — what a run with --no-cpu prices its kernels with, instead of another scene's counts."""
    if tag != "kitchen":
        return None
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_bench_default.json") or f.endswith("_bench_default_plain.json")), reverse=True):
        try:
            d = json.load(open(os.path.join(ROOT, "profiles", name)))
            oc = d["roofline"]["operation_counts"]
            if "cpu_baseline" in d and d["config"].get("headline_scene"):
                return (oc["lookups_per_ray"], oc["dtree_sample_levels_per_ray"], oc["dtree_pdf_levels_per_ray"]), name
        except Exception:
            pass
    return None


def algorithmic_bytes(work=None, rays=None, bvh=None, tag=None, sort_in_trace=False):
    """ALGORITHMIC bytes per unit of each path kernel: bytes the kernel's algorithm touches per unit, cache-oblivious,
    from the data layout of ppg_kernels.h and operation counts of the same workload: `work` = ppgo_work_counters of the CPU restatement
    with `rays` rays traced (S-tree lookups, D-tree levels), `bvh` = (nodes visited, triangles tested, rays) counted by k_trace itself."""
    source = "this run's cpu_baseline (ppgo_work_counters of the CPU restatement on the same scene and settings)"
    if work and rays:
        lookups, ds, dp = work[1] / rays, work[2] / rays, work[4] / rays
    else:
        stored = stored_operation_counts(tag)
        if stored:
            (lookups, ds, dp), source = stored[0], "profiles/%s (the newest committed line of this workload that ran the CPU restatement; this run had --no-cpu)" % stored[1]
        else:
            lookups, ds, dp, source = 0.70, 1.31, 2.48, "constants measured on cbox-720p (no CPU leg in this run and no committed line of this workload)"
    state_rd = 4 + 5 * 16                              # queue entry, ray_d, thr, li, hit, misc
    state_wr = 5 * 16 * lookups                        # ray_o, ray_d, thr, li (per surviving path) + misc
    shade = (state_rd + state_wr + 48 + 16             # + hit triangle (3 float4) + material
             + lookups * (4 + 64)                      # grid cell + leaf header
             + (ds + dp) * 32                          # sampling-tree nodes
             + lookups * 64)                           # speculative vertex record (4 float4)
    trace = 32 + 16                                    # ray in, hit out (small scenes: the triangles are LDS resident)
    detail = {"lookups_per_ray": lookups, "dtree_sample_levels_per_ray": ds, "dtree_pdf_levels_per_ray": dp, "source": source}
    if bvh and bvh[2]:
        n_bar, t_bar = bvh[0] / bvh[2], bvh[1] / bvh[2]
        trace += n_bar * 64 + t_bar * 48               # quantised BVH4 node = 64 B, TriAccel record = 48 B
        detail.update(bvh4_nodes_per_ray=n_bar, triangles_tested_per_ray=t_bar)
    if sort_in_trace:                                  # k_trace also sorts its queue slices by BSDF type (sort_slice):
        trace += 4 + 16 + 16 + 4 + 1                   # hit word, triangle word, material words read; sorted index + key byte written
        detail.update(slice_sort_inside_k_trace=True)
    # the commit, per RECORDED VERTEX: k_commit reads the path word + the vertex slot (4 float4; 6 with a spatial filter) and adds to ~5 leaf
    # accumulators + the weight; a round of the optimiser instead writes records (k_commit_records: path word amortised over the path's
    # vertices + 6 float4 read, key 8 + optimiser record 32 + splat record 16 written), sorts them, and k_splat_sorted reads key 8 + index 4
    # + splat record 16 per record (the D-tree is in LDS); k_adam_apply reads key 8 + index 4 + record 32
    return {"k_shade": shade, "k_trace": trace, "k_tail": trace + shade, "k_commit": 16 + 64 + 5 * 8 + 16, "k_commit_records": 4 + 96 + 56, "k_splat_sorted": 28,
            "k_adam_apply": 44, "k_generate": 80, "k_film": 4 * 16 + 88, "detail": detail}


def measured_traffic(tag):
    """HBM bytes per unit from profiles/<tag>_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, calibrated as
    MI355X_MICROARCH.md prescribes; written by tools/collect_profiles.py on the GPU box) — a STORED calibration, not measured in this run."""
    for name in ("r06_pmc_traffic_%s.json" % tag, "r05_pmc_traffic_%s.json" % tag, "r04_pmc_traffic_%s.json" % tag, "r03_pmc_traffic_%s.json" % tag, "r02_pmc_traffic_%s.json" % tag, "r01_pmc_traffic.json" if tag == "cbox" else ""):
        p = os.path.join(ROOT, "profiles", name)
        if name and os.path.exists(p):
            try:
                return json.load(open(p)), name
            except Exception:
                pass
    return None, None


def scene_file_check(path):
    """sha256 of a flat scene file against scratch/SHA256SUMS (tracked; tools/make_scenes.sh regenerates the files from the reference
    checkout and verifies them): the benchmark's workload is an untracked conversion of the reference's scene data, so the line says
    which bytes it rendered."""
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    digest, expected = h.hexdigest(), None
    sums = os.path.join(os.path.dirname(path), "SHA256SUMS")
    if os.path.exists(sums):
        for line in open(sums):
            f_ = line.split()
            if len(f_) == 2 and f_[1] == os.path.basename(path):
                expected = f_[0]
    return {"file": os.path.relpath(path, ROOT), "sha256": digest, "matches_SHA256SUMS": (digest == expected) if expected else None}


def scene_props(path, base):
    props = dict(base)
    pf = path + ".props"
    if os.path.exists(pf):
        for line in open(pf):
            if "=" in line:
                k, v = line.strip().split("=", 1)
                if k not in ("budget", "budgetType"):
                    props[k] = int(v) if v.lstrip("-").isdigit() else (float(v) if v.replace(".", "", 1).replace("-", "", 1).isdigit() else v)
    return props


def rmse(a, b):
    import numpy as np
    d = (np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2
    return float(np.sqrt(np.nanmean(d)))


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
        if args.force_dist and args.gpus == 1:  # stand-alone plumbing check: a one-rank communicator without a launcher
            for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
                os.environ.setdefault(k, v)
        dist.init_process_group("nccl")
        assert dist.get_world_size() == args.gpus, "launch with --nproc-per-node == --gpus"
        world = dist.get_world_size()
    import torch  # noqa: F811  (device sync + barrier plumbing only)

    base = dict(budgetType="spp", seed=1234, device=local_rank)
    scene_name = args.scene
    if args.scene_file:
        scene_name = "file"
    elif scene_name == "kitchen" and not os.path.exists(KITCHEN_FILE):
        scene_name = "room"
        if rank == 0:
            print("bench.py: scratch/kitchen-improved.ppgs is MISSING (tools/make_scenes.sh makes it from the reference checkout): rendering the "
                  "procedural stand-in `room` instead - this is NOT the BASELINE.json configuration", file=sys.stderr, flush=True)
    scene_check = None
    traffic_tag = "cbox" if scene_name == "cbox" else "kitchen"
    if scene_name in ("kitchen", "file"):
        path = args.scene_file or KITCHEN_FILE
        scene = ppg_host.load_scene_file(path)
        scene_check = scene_file_check(path)
        if scene_check["matches_SHA256SUMS"] is False:
            raise SystemExit("bench.py: %s does not match scratch/SHA256SUMS (regenerate it with tools/make_scenes.sh)" % path)
        if args.size_override:
            scene.camera = ppg_host.resize_camera(scene.camera, args.width, args.height)  # keeps the horizontal field of view
        if args.constant_env:
            scene.environment = tuple(float(v) for v in args.constant_env.split(","))
        args.width, args.height = scene.camera["width"], scene.camera["height"]
        if args.all_diffuse:  # experiment: what the BSDF mix costs — every surface a grey two-sided Lambertian, no textures
            scene.materials = [dict(type=1, reflectance=(0.5, 0.5, 0.5)) for _ in scene.materials]
        props = scene_props(path, base)
        spp = int(props.get("sppPerPass", 4))
        if scene_name == "kitchen":
            workload = ("kitchen-improved-720p: the reference's scenes/kitchen/kitchen-improved.xml converted in the development container (%d triangles: 283 of its 289 "
                        "meshes, six are missing from the reference checkout; %d BSDFs, %d bitmap textures, sunsky baked into a %dx%d radiance map), %dx%d, "
                        "improved preset (inversevar, kl, stochastic + box filters, sTreeThreshold 4000), %d spp/pass, maxDepth %d, %d passes"
                        % (scene.n_triangles, len(scene.materials), len(scene.textures), scene.envmap["rgb"].shape[1] if scene.envmap else 0,
                           scene.envmap["rgb"].shape[0] if scene.envmap else 0, args.width, args.height, spp, int(props.get("maxDepth", -1)), args.steps))
        else:
            workload = "%s (%d triangles, %d spheres), %dx%d, %d spp/pass, %d passes, the scene file's integrator settings" % (
                os.path.basename(path), scene.n_triangles, len(scene.spheres), args.width, args.height, spp, args.steps)
    elif scene_name == "torus":
        # BASELINE.json configs[4] "TORUS (SDS caustics), 1920x1080, sppPerPass=1, sTreeThreshold=4000": the paper's scene is not bundled with
        # the reference — the labelled procedural stand-in of SURVEY.md §8(d) S5 (diffuse torus in a glass cube, one small emitter)
        if (args.width, args.height) == (1280, 720):
            args.width, args.height = 1920, 1080
        spp = 1
        scene = ppg_host.torus_scene(args.width, args.height)
        props = dict(base, sppPerPass=1, sTreeThreshold=4000, maxDepth=-1, rrDepth=5, strictNormals=0)
        workload = "torus-1080p (torus-class STAND-IN: diffuse torus in a glass cube, %d triangles), %dx%d, 1 spp/pass, %d passes, sTreeThreshold 4000" % (
            scene.n_triangles, args.width, args.height, args.steps)
    elif scene_name == "cbox":
        spp = args.spp or 4
        scene = ppg_host.cbox_scene(args.width, args.height)
        props = dict(base, sppPerPass=spp, maxDepth=10, rrDepth=10, strictNormals=1)
        workload = "cbox-720p: procedural CBOX (36 tris), %dx%d, %d spp/pass, %d passes, default SD-tree params, maxDepth 10" % (args.width, args.height, spp, args.steps)
    else:
        # the bundled KITCHEN cannot be read here (no scratch/kitchen-improved.ppgs): labelled procedural stand-in of SURVEY.md §8(d) S3 with the
        # README's "improved" preset; maxDepth -1 / rrDepth 5 as in kitchen-improved.xml
        spp = args.spp or 1
        scene = ppg_host.room_scene(args.width, args.height, n_boxes=args.room_boxes, tess=8, glossy=args.glossy)
        props = dict(base, **IMPROVED)
        props.update(sppPerPass=spp, maxDepth=-1, rrDepth=5, strictNormals=0)
        workload = "room-720p (kitchen-class STAND-IN for the KITCHEN scene, %d %s triangles), %dx%d, %d spp/pass, %d passes, improved preset" % (
            scene.n_triangles, "Lambertian / GGX(0.1) / plastic" if args.glossy else "Lambertian", args.width, args.height, spp, args.steps)

    def make(budget_passes, timing=False, the_scene=scene, the_props=props, the_spp=None, seed=None):
        p = dict(the_props)
        if seed is not None:
            p["seed"] = seed
        e = ppg_host.Engine.hip(budget=float(budget_passes * (the_spp or spp)), **p)
        e.set_scene(the_scene)
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

    def timed_render(gpt):
        sync()
        t0 = time.perf_counter()
        img = gpt.render()
        sync()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return img, dt

    if args.warmup > 0:
        w_ = make(args.warmup)
        w_.render()
        w_.engine.close()
        del w_
    # `value` is the MEDIAN of --repeats complete renders (new context each: nothing carries over but the process-wide block cache); boxes and
    # individual renders differ by more than a kernel change is worth (DESIGN.md §7 "Boxes differ"), min / max are reported beside it
    # (one context at a time: a context of this workload holds ~50 GB of path state and vertex slots — five of them alive at once ran the
    # later renders at a third of the speed)
    runs = []
    for k_ in range(max(1, args.repeats)):
        if os.environ.get("PPG_DEBUG_ALLOC"):
            print("[bench] timed render %d" % k_, file=sys.stderr, flush=True)
        g_ = make(args.steps)
        gc.collect()  # (a full collection of the interpreter's heap takes tens of ms with torch loaded: not inside a 120 ms render)
        _, dt_ = timed_render(g_)
        runs.append((dt_, g_.iterations))
        g_.engine.close()  # (not left to the garbage collector: the fifth render of five used to run beside the contexts of the four before it)
        del g_
    order = sorted(range(len(runs)), key=lambda k: runs[k][0])
    dt, iterations = runs[order[len(order) // 2]]
    all_dt = [r[0] for r in runs]
    del runs
    samples = args.width * args.height * spp * args.steps
    rays = sum(s["rays"] for it in iterations for s in it["stats"])
    own_samples = sum(s["samples"] for it in iterations for s in it["stats"])
    plen = sum(s["path_length_sum"] for it in iterations for s in it["stats"])
    var_last = iterations[-1]["stats"][-1]["variance"]
    out = {
        "metric": "Msamples/s", "value": samples / dt / 1e6, "unit": "Msamples/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "real-scene" if scene_name in ("kitchen", "file") else "synthetic",  # (the reference's own scene data, converted; camera rays and samples are generated)
        "config": {"workload": workload, "scene_file": scene_check, "headline_scene": scene_name == "kitchen",
                   "iterations": [it["passes"] for it in iterations], "parallelism": "tiles%d" % args.gpus,
                   "rays_per_sample": rays / max(1, own_samples), "avg_path_length": plen / max(1, own_samples), "variance_last_iteration": var_last},
        "tuning_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("PPG_")},
        "repeats": {"n": len(all_dt), "value_is": "median", "values": [samples / t / 1e6 for t in all_dt],
                    "min": samples / max(all_dt) / 1e6, "max": samples / min(all_dt) / 1e6},
    }
    if scene_name == "kitchen":
        out["reference_log"] = dict(REF_KITCHEN_LOG, triangle_ratio_of_this_scene=scene.n_triangles / REF_KITCHEN_LOG["primitives"],
                                    note="the reference binary on the complete scene (1 414 390 primitives) at 700x400; this run renders the %d triangles of the "
                                         "checkout (six meshes are missing from it) at %dx%d" % (scene.n_triangles, args.width, args.height))
        out["vs_reference_log"] = out["value"] / REF_KITCHEN_LOG["msamples_per_s"]
    if rank == 0 and args.gpus == 1 and not args.no_single_call:
        # the same render through ppg_render(): one C-ABI call, the iteration loop in the library (what the C++ host / the plug-in make)
        e1 = ppg_host.Engine.hip(budget=float(args.steps * spp), **props)
        e1.set_scene(scene)
        sync()
        t1 = time.perf_counter()
        e1.render()
        sync()
        dt1 = time.perf_counter() - t1
        out["single_call"] = {"entry_point": "ppg_render()", "value": samples / dt1 / 1e6, "unit": "Msamples/s", "seconds": dt1}
        del e1

    work = rays_cpu = None
    if rank == 0 and args.gpus == 1 and not args.no_cpu:
        # CPU baseline: the oracle (a port, not the reference binary) on the host cores, bounded sample:
        # the same scene / resolution / settings, the first `cpu_passes` passes of the same schedule.
        import ctypes
        # (threads: the port does not scale beyond a few dozen — measured on the MI355X box's 256 hardware threads, 7 passes of this render: 0.461
        # Msamples/s with 256 threads, 0.680 with 128, 0.821 with 64, 0.891 with 32 (profiles/r06_experiments.json) — so it runs with the count
        # that is fastest there, and `cores` says which)
        cores = min(os.cpu_count() or 1, args.cpu_threads) if args.cpu_threads > 0 else (os.cpu_count() or 1)
        lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libppg_oracle.so"))
        cp = args.cpu_passes if args.cpu_passes > 0 else min(args.steps, 20)  # (the GPU's own pass schedule when the render is short enough)
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
        # ... and ONE core, for per-core normalisation: the same scene and settings on a film shrunk to 1/8 x 1/8 (same field of
        # view), 3 passes — a whole film on one core would take minutes per pass
        one = None
        if hasattr(scene, "camera") and not args.no_cpu_1core:
            import copy
            small1 = copy.copy(scene)
            w1, h1 = max(16, args.width // 8), max(9, args.height // 8)
            small1.camera = ppg_host.resize_camera(scene.camera, w1, h1)
            o1 = ppg_host.Engine(lib, "ppgo_", budget=float(3 * spp), **{k: v for k, v in props.items() if k != "device"})
            lib.ppgo_set_modes(o1.ctx, 0, 0, 1)
            o1.set_scene(small1)
            t1 = time.perf_counter()
            ppg_host.GuidedPathTracer(engine=o1).render()
            dt1c = time.perf_counter() - t1
            one = {"value": w1 * h1 * spp * 3 / dt1c / 1e6, "unit": "Msamples/s", "cores": 1, "seconds": dt1c,
                   "sample": "3 passes of the same scene and settings on a %dx%d film (the full film on one core takes minutes per pass)" % (w1, h1)}
            del o1
        out["cpu_baseline"] = {"value": args.width * args.height * spp * cp / dtc / 1e6, "unit": "Msamples/s", "cores": cores,
                               "kind": "port", "sample": "first %d passes (%d spp) of the same render(), oracle restatement, OpenMP over 32x32 blocks"
                               % (cp, cp * spp), "seconds": dtc, "one_core": one,
                               "host_hardware_threads": os.cpu_count(),
                               "note": "a PORT on this box's host cores (brute-force / median-split BVH, software libm: slower than the reference binary was on 16 "
                                       "cores), run with the thread count at which it is fastest on this box (--cpu-threads; all 256 hardware threads give half of "
                                       "it) — a reported baseline, not the yardstick; the reference's own figure is `reference_log`"}
        del og, o

    times = None
    if not args.no_roofline:
        # separate instrumented render: per-kernel HIP-event durations on the kernels' stream.  EVERY rank runs it (the render
        # contains collectives); rank 0 reports its own kernels.
        g2 = make(args.steps, timing=True)
        g2.render()
        times = g2.engine.kernel_times()
        rays2 = sum(s["rays"] for it in g2.iterations for s in it["stats"])
        its2 = g2.iterations
        sync()
        del g2
    if rank == 0 and times:
        counts = {k["name"]: k["units"] for k in times if k["launches"] == 0}
        times = [k for k in times if k["launches"] > 0]
        trace_rays = sum(k["units"] for k in times if k["name"] == "k_trace")
        bvh = (counts.get("bvh_nodes_visited", 0), counts.get("bvh_triangles_tested", 0), trace_rays) if counts else None
        # (FULL scenes sort every bounce's queue slices by BSDF type: inside k_trace since round 6, unless PPG_SORT_KERNEL=1 launches k_sort_slices)
        sort_in_trace = any(k["name"].startswith("k_shade<common>") for k in times) and not any(k["name"] == "k_sort_slices" for k in times)
        alg = algorithmic_bytes(work, rays_cpu, bvh, traffic_tag, sort_in_trace)
        # the commit's unit is a recorded vertex (the library books its launches by paths): iterations rendered in rounds of the optimiser
        # commit through k_commit_records / k_splat_sorted, the others (the first one; every one without a learned fraction) through k_commit
        rounds_on = props.get("bsdfSamplingFractionLoss", "none") != "none" and props.get("spatialFilter", "nearest") != "box" and props.get("nee", "never") != "kickstart"
        v_round = sum(s["vertices_committed"] for it in its2 for s in it["stats"] if rounds_on and it["iter"] > 0)
        v_plain = sum(s["vertices_committed"] for it in its2 for s in it["stats"]) - v_round
        commit_units = {"k_commit": v_plain, "k_commit_records": v_round, "k_splat_sorted": v_round}
        tr, tr_file = measured_traffic(traffic_tag)

        def entry(k):
            name = k["name"].split("<")[0]
            units = k["units"]
            if name == "k_tail":  # the persistent-thread tail traces AND shades: its unit is a ray it traced = all rays - those k_trace traced
                units = max(0, rays2 - trace_rays)
            if name in commit_units:
                units = commit_units[name]
            bpu = alg.get(name)
            if bpu is None or not units:
                return None
            avg_units, avg_ms = units / k["launches"], k["ms"] / k["launches"]
            achieved = bpu * avg_units / (avg_ms * 1e-3) / 1e9
            traffic = None
            tkey = k["name"] if tr and k["name"] in tr.get("bytes_per_unit", {}) else name   # (r04: k_shade<common> / k_shade<rest> on their own)
            if tr and tkey in tr.get("bytes_per_unit", {}):  # the stored calibration's unit (r03 on: the roofline's own unit for every kernel; r02: k_tail per path handed over)
                per_ray = "unit_of_work" in tr
                traffic = tr["bytes_per_unit"][tkey] * (units if (per_ray or name != "k_tail") else k["units"]) / k["launches"]
            return {"kernel": k["name"], "ms": round(k["ms"], 3), "launches": k["launches"], "avg_launch_ms": avg_ms, "avg_units_per_launch": avg_units,
                    "algorithmic_bytes_per_unit": bpu, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "traffic": traffic}

        per = {k["name"]: entry(k) for k in times}
        per = {n: e for n, e in per.items() if e}
        dom = max(per.values(), key=lambda e: e["ms"])
        name = dom["kernel"].split("<")[0]
        out["roofline"] = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": dom["frac"], "traffic": dom["traffic"],
                           "traffic_source": ("stored calibration profiles/%s (PMC bytes per unit of an earlier run) x this run's units" % tr_file) if dom["traffic"] is not None else None,
                           "avg_launch_ms": dom["avg_launch_ms"], "launches": dom["launches"],
                           "algorithmic_bytes_per_unit": dom["algorithmic_bytes_per_unit"], "avg_units_per_launch": dom["avg_units_per_launch"],
                           "unit_of_work": "ray traced and shaded inside the persistent-thread tail" if name == "k_tail" else ("traced ray" if name in ("k_trace", "k_shade") else "unit of " + name),
                           "units_of_the_commit": "recorded vertex (k_commit, k_commit_records, k_splat_sorted); optimiser record (k_adam_apply)",
                           "tail_critical_path": ({"longest_paths_sum_bounces": counts["tail_longest_paths_sum"], "k_tail_ms": next((k["ms"] for k in times if k["name"] == "k_tail"), None),
                                                   "us_per_bounce_of_the_longest_path": 1e3 * next((k["ms"] for k in times if k["name"] == "k_tail"), 0.0) / counts["tail_longest_paths_sum"],
                                                   "note": "a launch of k_tail cannot end before its longest path has: the sum over its launches of the longest path each finished "
                                                           "(bounces, counted from the camera: the first ones ran in the wavefront) against k_tail's total time — the average time per "
                                                           "bounce of the lane that decides when a launch ends, crowd phase included.  A large batch's k_tail is bound by this chain of "
                                                           "dependent bounces, the k_tail of a small training round by the throughput of its crowded waves (DESIGN.md section 7) - "
                                                           "neither by bandwidth"}
                                                  if counts.get("tail_longest_paths_sum") else None),
                           "operation_counts": alg["detail"], "kernels_ms": {k["name"]: round(k["ms"], 3) for k in times},
                           "per_kernel": {n: {q: (round(v, 4) if isinstance(v, float) else v) for q, v in e.items() if q != "kernel"} for n, e in per.items()
                                          if n.split("<")[0] in ("k_trace", "k_shade", "k_tail", "k_commit", "k_commit_records", "k_splat_sorted", "k_adam_apply")},
                           "note": "the BVH and the SD-tree are L2 / Infinity-Cache resident: the algorithmic bytes are what the kernel must read per ray "
                                   "(cache-oblivious), `traffic` what reaches HBM.  k_tail finishes the paths still alive after the wavefront bounces, one lane per "
                                   "path: it is bound by the chain of dependent bounces of its longest path (`tail_critical_path`), not by bandwidth (DESIGN.md §7)"}

    if rank == 0 and args.gpus == 1 and not args.no_rmse and scene_name == "kitchen" and os.path.exists(KITCHEN_REFERENCE):
        # Time to equal error.  The yardstick is the reference's own converged picture, scenes/kitchen/kitchen-reference.exr (committed as
        # a fixture: pixels are data), so the scene is rendered at the reference's film size, 700x400.  Target: the error of the
        # reference's own guided render kitchen-improved.exr (2400 spp, 500.9 s on 16 CPU cores) against that picture, over the pixels
        # outside the footprint of the six meshes missing from the reference checkout (tools/make_kitchen_mask.py; 21 % masked).
        import copy
        fx = np.load(KITCHEN_REFERENCE)
        ref = fx["rgb"].astype(np.float64)
        blk = int(fx["mask_block"])
        keep = ~np.kron(fx["mask_blocks"], np.ones((blk, blk), np.uint8)).astype(bool)
        target_mape, target_rmse = float(fx["kitchen_improved_mape_unmasked"]), float(fx["kitchen_improved_rmse_unmasked"])
        small = copy.copy(scene)
        small.camera = ppg_host.resize_camera(scene.camera, ref.shape[1], ref.shape[0])
        make(3, the_scene=small).render()
        trials = []

        def trial(n):
            img, t = timed_render(make(n, the_scene=small, seed=4321))
            d = (np.asarray(img, np.float64) - ref)[keep]
            trials.append({"spp": n * spp, "seconds": t, "mape": float((np.abs(d) / (ref[keep] + 0.01)).mean()), "rmse": float(np.sqrt((d * d).mean()))})
            return trials[-1]

        a, b = trial(255), trial(1023)
        # error ~ spp^-slope between the two; aim 4 % under the target, confirm by rendering that budget (and once more if it misses)
        slope = max(0.2, math.log(a["mape"] / b["mape"]) / math.log(b["spp"] / a["spp"]))
        n = b["spp"]
        for _ in range(2):
            if trials[-1]["mape"] <= target_mape:
                break
            n = int(min(4800, max(n + 1, n * (trials[-1]["mape"] / (0.96 * target_mape)) ** (1.0 / slope))))
            trial(-(-n // spp))
        hit_mape = min((t for t in trials if t["mape"] <= target_mape), key=lambda t: t["seconds"], default=None)
        hit_rmse = min((t for t in trials if t["rmse"] <= target_rmse), key=lambda t: t["seconds"], default=None)
        # the budget that met the MAPE target, again with other seeds: how much of the figure is one seed's luck
        seeds = []
        all_seeds = None  # the budget at which EVERY seed meets the target, and its slowest render: the figure that does not depend on one seed's luck
        if hit_mape:
            def seed_runs(n_spp, first=None):
                runs_ = []
                for sd in (4321, 99, 20260927):
                    if first is not None and sd == 4321:
                        runs_.append({"seed": sd, "seconds": first["seconds"], "mape": first["mape"], "rmse": first["rmse"]})
                        continue
                    img, t = timed_render(make(-(-n_spp // spp), the_scene=small, seed=sd))
                    d = (np.asarray(img, np.float64) - ref)[keep]
                    runs_.append({"seed": sd, "seconds": t, "mape": float((np.abs(d) / (ref[keep] + 0.01)).mean()), "rmse": float(np.sqrt((d * d).mean()))})
                return runs_
            seeds = seed_runs(hit_mape["spp"], hit_mape)
            n_all, runs_all = hit_mape["spp"], seeds
            for _ in range(3):
                worst = max(r_["mape"] for r_ in runs_all)
                if worst <= target_mape:
                    break
                n_all = int(min(6000, max(n_all + 16, n_all * (worst / (0.97 * target_mape)) ** (1.0 / slope))))
                runs_all = seed_runs(n_all)
            if max(r_["mape"] for r_ in runs_all) <= target_mape:
                all_seeds = {"spp": n_all, "seconds_slowest_seed": max(r_["seconds"] for r_ in runs_all), "runs": runs_all,
                             "speedup_vs_reference_log": REF_KITCHEN_SECONDS / max(r_["seconds"] for r_ in runs_all)}
        # The same scene with the DEFAULT settings of scenes/kitchen/kitchen.xml (4 spp per pass, nearest filters, no learned fraction, automatic
        # sample combination) at the reference's own budget, 2400 spp: this build's error against the converged picture next to the error of
        # the reference's own render of that configuration (kitchen.exr), and its time next to that render's log (884.6 s on 12 CPU cores).
        default_preset = None
        try:
            dprops = dict(base, strictNormals=1)
            make(7, the_scene=small, the_props=dprops, the_spp=4).render()
            dimg, dt = timed_render(make(600, the_scene=small, the_props=dprops, the_spp=4, seed=4321))
            dd = (np.asarray(dimg, np.float64) - ref)[keep]
            default_preset = {"settings": "scenes/kitchen/kitchen.xml: strictNormals, 2400 spp; everything else the plug-in's defaults", "spp": 2400, "seconds": dt,
                              "mape": float((np.abs(dd) / (ref[keep] + 0.01)).mean()), "rmse": float(np.sqrt((dd * dd).mean())),
                              "reference_own_render": {"file": "scenes/kitchen/kitchen.exr", "mape": float(fx["kitchen_mape_unmasked"]), "rmse": float(fx["kitchen_rmse_unmasked"]),
                                                       "seconds": 884.562, "cores": 12, "source": "its embedded log (tests/golden/ref_logs.json)"},
                              "speedup_vs_reference_log_at_equal_spp": 884.562 / dt}
        except Exception as ex:  # (the leg is a by-product: it must not cost the line)
            default_preset = {"error": str(ex)}
        out["time_to_rmse"] = {
            "reference_image": "scenes/kitchen/kitchen-reference.exr of the reference (tests/golden/ref_kitchen_reference.npz), 700x400; %.0f %% of the film "
                               "masked: footprint of the 6 meshes missing from the reference checkout" % (100 * (1 - keep.mean())),
            "target_source": "error of the reference's own kitchen-improved.exr (2400 spp, 500.9 s on 16 CPU cores, its embedded log) against the same picture, same pixels",
            "target_mape": target_mape, "target_rmse": target_rmse,
            "seconds_to_mape": hit_mape["seconds"] if hit_mape else None, "spp_to_mape": hit_mape["spp"] if hit_mape else None,
            "seconds_to_rmse": hit_rmse["seconds"] if hit_rmse else None, "spp_to_rmse": hit_rmse["spp"] if hit_rmse else None,
            "speedup_vs_reference_log_mape": (REF_KITCHEN_SECONDS / hit_mape["seconds"]) if hit_mape else None,
            "speedup_vs_reference_log_rmse": (REF_KITCHEN_SECONDS / hit_rmse["seconds"]) if hit_rmse else None,
            "seeds_at_spp_to_mape": seeds, "all_seeds_meet_target": all_seeds,
            "seconds_to_mape_min_max": [min(s_["seconds"] for s_ in seeds), max(s_["seconds"] for s_ in seeds)] if seeds else None,
            "mape_min_max": [min(s_["mape"] for s_ in seeds), max(s_["mape"] for s_ in seeds)] if seeds else None,
            "seeds_meeting_target": sum(1 for s_ in seeds if s_["mape"] <= target_mape) if seeds else None,
            "trials": trials,
            "default_preset": default_preset,
            "note": "MAPE = mean |x - ref| / (ref + 0.01) is the meaningful figure: the reference render's RMSE is a handful of fireflies (max pixel 119), "
                    "which this build's renders of equal spp do not show to that extent (seed dependent), so the RMSE target is met at well under half the samples.  Seconds = render() "
                    "of the 700x400 film, scene upload and BVH build excluded like the reference's kd-tree build"}

    if rank == 0 and args.gpus == 1 and not args.no_secondary and scene_name != "cbox":
        cb = ppg_host.cbox_scene(1280, 720)
        cprops = dict(base, sppPerPass=4, maxDepth=10, rrDepth=10, strictNormals=1)
        make(3, the_scene=cb, the_props=cprops, the_spp=4).render()
        _, t = timed_render(make(args.secondary_passes, the_scene=cb, the_props=cprops, the_spp=4))
        out["secondary"] = {"workload": "cbox-720p (BASELINE.json configs[1]): procedural CBOX, 1280x720, 4 spp/pass, %d passes, default SD-tree params, maxDepth 10" % args.secondary_passes,
                            "value": 1280 * 720 * 4 * args.secondary_passes / t / 1e6, "unit": "Msamples/s"}

    if dist is not None:
        dist.barrier()
    if rank == 0:  # the one JSON line is the last thing written
        sys.stdout.flush(); sys.stderr.flush()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--cpu-threads", type=int, default=32, help="OpenMP threads of the cpu_baseline leg (0 = all hardware threads)")
    ap.add_argument("--steps", type=int, default=127, help="render passes in the timed render() (budget = steps * spp)")
    ap.add_argument("--warmup", type=int, default=3, help="passes of the throw-away warm-up render")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel and pass of the procedural scenes (default: 4 cbox, 1 room / torus); scene files carry their own")
    ap.add_argument("--scene", choices=["kitchen", "cbox", "room", "torus"], default="kitchen",
                    help="kitchen = the reference's KITCHEN scene, improved preset (scratch/kitchen-improved.ppgs; falls back to `room`); room = kitchen-class procedural "
                         "stand-in, improved preset; cbox = BASELINE configs[1]; torus = torus-class stand-in (SDS caustics)")
    ap.add_argument("--scene-file", help="flat scene file (ppg_host.save_scene / `python -m ppg_host scene.xml --ppgs`) instead of a named scene")
    ap.add_argument("--size-override", action="store_true", help="with a scene file: render at --width x --height instead of the file's film size")
    ap.add_argument("--constant-env", help="with a scene file: R,G,B of a constant environment emitter (STAND-IN lighting)")
    ap.add_argument("--room-boxes", type=int, default=1820, help="boxes of the room scene (768 triangles each)")
    ap.add_argument("--glossy", action="store_true", help="room scene with the S3 material mix (GGX alpha 0.1 metal, plastic) instead of Lambertian only")
    ap.add_argument("--cpu-passes", type=int, default=0, help="passes timed on the CPU baseline (bounded sample); 0 = min(--steps, 20): the GPU's own schedule for the driver's command")
    ap.add_argument("--no-cpu-1core", action="store_true", help="skip the one-core leg of the CPU baseline")
    ap.add_argument("--secondary-passes", type=int, default=63)
    ap.add_argument("--repeats", type=int, default=5, help="timed renders; value = their median (min / max reported)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-rmse", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-single-call", action="store_true", help="skip the extra render through ppg_render() (the single C-ABI call)")
    ap.add_argument("--all-diffuse", action="store_true", help="experiment: replace every BSDF of a scene file by a grey two-sided Lambertian")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the reducer even with one rank (plumbing check)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # the plain command `python bench.py --gpus N ...`: start the N ranks ourselves (one process per GPU over RCCL), as the launcher would
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    run(args)


if __name__ == "__main__":
    main()
