#!/usr/bin/env python3
"""GPU box, round 2: the rocprofv3 evidence behind bench.py's KITCHEN line → gpurun_out/profiles/r02_*.

Each item is its own rocprofv3 invocation (counters never share a run with tracing; FETCH_SIZE and WRITE_SIZE need separate passes:
MI355X_MICROARCH.md §rocprofv3 PMC slots):
  1. --kernel-trace --stats of EXACTLY the driver's command, `python bench.py --steps 20 --warmup 5`
        → r02_bench_default_kernel_stats.csv + the JSON line that run printed (r02_bench_default.json).  The instrumented render behind
          the `roofline` block is the only one that launches k_trace<false, true> (the variant that counts BVH visits), so that row's
          average duration is the figure to compare with roofline.avg_launch_ms.
  2. --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT TCC_MISS on one 20-pass render → r02_pmc_traffic_kitchen.json (HBM bytes per unit per
          kernel; read / write factors calibrated on k_film / k_generate, whose streamed bytes are known exactly)
  3. --pmc SQ_* wave-cycle breakdown of the same render, once with the shipped library and once with a build without the leaf vote
          (lib/libppg_hip_v0.so, -DPPG_LEAF_VOTE=0: the traversal before this round's last change) → r02_pmc_wait_cycles_k_trace.json
"""
import collections, csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(OUT, "profiles")
os.makedirs(PROF, exist_ok=True)
STEPS = 20
DRIVER = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5"]
ONE = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(STEPS), "--warmup", "0", "--no-cpu", "--no-rmse", "--no-secondary", "--no-roofline"]
env = dict(os.environ, TMPDIR="/tmp")


def prof(tag, args, cmd, extra_env=None, stdout=None):
    d = os.path.join(OUT, tag)
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3"] + args + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd, cwd="/tmp", env=dict(env, **(extra_env or {})),
                   stdout=stdout or subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    return d


def short(name):
    return name.split("(")[0].replace("void ", "").split("<")[0]


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    os.remove(f[0])
    return agg


what = sys.argv[1:] or ["stats", "traffic", "wait"]

if "stats" in what:
    with open(os.path.join(PROF, "r02_bench_default.json"), "w") as fo:
        d1 = prof("prof_stats", ["--kernel-trace", "--stats"], DRIVER, stdout=fo)
    for f in glob.glob(os.path.join(d1, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(PROF, "r02_bench_default_kernel_stats.csv"))
    shutil.rmtree(d1, ignore_errors=True)

units = launches = None
if "traffic" in what or "wait" in what:
    sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))
    sys.path.insert(0, ROOT)
    import torch  # noqa: F401,E402
    import ppg_host  # noqa: E402
    from bench import KITCHEN_FILE, scene_props  # noqa: E402
    scene = ppg_host.load_scene_file(KITCHEN_FILE)
    props = scene_props(KITCHEN_FILE, dict(budgetType="spp", seed=1234))
    SPP = int(props.get("sppPerPass", 1))
    W, H = scene.camera["width"], scene.camera["height"]
    e = ppg_host.Engine.hip(budget=float(STEPS * SPP), **props)
    e.set_scene(scene); e.enable_kernel_timing(True)
    g = ppg_host.GuidedPathTracer(engine=e); g.render()
    units = collections.defaultdict(float); launches = collections.defaultdict(float)
    for k in e.kernel_times():
        units[k["name"].split("<")[0]] += k["units"]; launches[k["name"].split("<")[0]] += k["launches"]
    units["k_commit"] = sum(s["vertices_committed"] for it in g.iterations for s in it["stats"])  # unit of k_commit = recorded vertex
    del g, e

if "traffic" in what:
    a2 = counters(prof("pmc_fetch", ["--pmc", "FETCH_SIZE"], ONE))
    a3 = counters(prof("pmc_write", ["--pmc", "WRITE_SIZE", "TCC_HIT", "TCC_MISS"], ONE))
    KB = 1024.0
    gen_known_wr = 80.0 * units["k_generate"]                                           # ray_o, ray_d, thr, li, misc
    film_known_rd = 16.0 * W * H * SPP * STEPS + (4 + 11 * 4) * units["k_film"]          # every li sample once + per launch and pixel: index + 11 accumulators
    wr_factor = gen_known_wr / (a3["k_generate"]["WRITE_SIZE"] * KB) if a3["k_generate"]["WRITE_SIZE"] else None
    rd_factor = film_known_rd / (a2["k_film"]["FETCH_SIZE"] * KB) if a2["k_film"]["FETCH_SIZE"] else None
    res = {"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE TCC_HIT TCC_MISS, one %d-pass render of the bench workload (kitchen-improved-720p)" % STEPS,
           "calibration": {"read_factor_from_k_film": rd_factor, "write_factor_from_k_generate": wr_factor,
                           "note": "factor = known streamed bytes / (counter * 1024); MI355X_MICROARCH.md §HBM expects ~2 for reads"},
           "bytes_per_unit": {}, "per_kernel": {}}
    for k in sorted(set(a2) | set(a3)):
        rd = a2[k].get("FETCH_SIZE", 0.0) * KB * (rd_factor or 2.0)
        wr = a3[k].get("WRITE_SIZE", 0.0) * KB * (wr_factor or 1.0)
        hit, miss = a3[k].get("TCC_HIT", 0.0), a3[k].get("TCC_MISS", 0.0)
        u = units.get(k)
        res["per_kernel"][k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "l2_hit_rate": hit / (hit + miss) if hit + miss else None, "units": u, "launches": launches.get(k)}
        if u:
            res["bytes_per_unit"][k] = (rd + wr) / u
    json.dump(res, open(os.path.join(PROF, "r02_pmc_traffic_kitchen.json"), "w"), indent=1)
    print(json.dumps(res["bytes_per_unit"], indent=1))

if "wait" in what:
    SQ = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"]
    res = {"source": "rocprofv3 --pmc " + " ".join(SQ) + ", one %d-pass render of kitchen-improved-720p; quad-cycle units" % STEPS, "variants": {}}
    v0 = os.path.join(ROOT, "practical-path-guiding_amd", "lib", "libppg_hip_v0.so")
    for tag, ex in (("leaf_vote_16 (shipped)", {}), ("leaf_vote_off (before)", {"PPG_HIP_LIB": v0} if os.path.exists(v0) else None)):
        if ex is None:
            continue
        a = counters(prof("pmc_sq", ["--pmc"] + SQ, ONE, ex))
        out = {}
        for k in ("k_trace", "k_shade", "k_tail", "k_commit"):
            c = a.get(k)
            if not c:
                continue
            wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            out[k] = dict(c, wait_any_share=c.get("SQ_WAIT_ANY", 0.0) / wc, wait_inst_share=c.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                          active_share=c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                          valu_lane_utilisation=None, rays=units.get(k) if k in ("k_trace", "k_shade") else None)
        res["variants"][tag] = out
    json.dump(res, open(os.path.join(PROF, "r02_pmc_wait_cycles_k_trace.json"), "w"), indent=1)
    print(json.dumps({t: {k: {m: round(v[m], 3) for m in ("wait_any_share", "wait_inst_share", "active_share")} for k, v in o.items()} for t, o in res["variants"].items()}, indent=1))
