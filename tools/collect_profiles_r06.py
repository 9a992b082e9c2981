#!/usr/bin/env python3
"""GPU box, round 6: the rocprofv3 evidence behind bench.py's KITCHEN line → gpurun_out/profiles/r06_* (copied into profiles/ afterwards).

Each item is its own rocprofv3 invocation (counters never share a run with tracing; FETCH_SIZE and WRITE_SIZE need separate passes:
MI355X_MICROARCH.md §rocprofv3 PMC slots):
  stats    --kernel-trace --stats of EXACTLY the driver's command, `python bench.py --steps 20 --warmup 5`
             → r06_bench_default.json (the JSON line that run printed), r06_bench_default_kernel_stats.csv (rocprofv3's summary: every render
               of the process), and — so that the roofline block can be recomputed from profiles/ alone — the launches of the INSTRUMENTED
               render only, cut out of the kernel trace (it is the render that launches k_trace<false, true>, the variant that counts BVH
               visits): r06_instrumented_render_launches.csv (one row per launch) + r06_instrumented_render_kernels.json (per kernel: launches,
               total and average duration — compare with roofline.per_kernel[*].avg_launch_ms of the JSON line)
  traffic  --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT TCC_MISS on one 20-pass render → r06_pmc_traffic_kitchen.json (HBM bytes per unit per
             kernel; read / write factors calibrated on k_film / k_generate, whose streamed bytes are known exactly)
  wait     --pmc SQ_* wave-cycle breakdown of the same render → r06_pmc_wait_cycles.json
"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(OUT, "profiles")
os.makedirs(PROF, exist_ok=True)
STEPS = 20
DRIVER = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5"]
ONE = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(STEPS), "--warmup", "0", "--no-cpu", "--no-rmse", "--no-secondary", "--no-roofline", "--no-single-call", "--repeats", "1"]
env = dict(os.environ, TMPDIR="/tmp")


def prof(tag, args, cmd, stdout=None):
    d = os.path.join(OUT, tag)
    shutil.rmtree(d, ignore_errors=True)
    subprocess.run(["rocprofv3"] + args + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd, cwd="/tmp", env=env,
                   stdout=stdout or subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    return d


def short(name):
    n = name.replace("void ", "")
    m = re.match(r"([A-Za-z_0-9:]+)(<[^(]*>)?", n)
    return (m.group(1).split("::")[-1] + (m.group(2) or "")) if m else n[:48]


def base(name):
    n = short(name)
    if n.startswith("k_shade<") and n.endswith(", true, 1>"):
        return "k_shade<common>"   # the common material classes (MSET_COMMON) ...
    if n.startswith("k_shade<false, false, true"):
        return "k_shade<rest>"     # ... and the complete kernel over the rest of the sorted slices
    if n.startswith("k_film_groups"):
        return "k_film"            # (the final iteration's film kernel: the same stream of li samples, booked under the same timer name)
    return n.split("<")[0]


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return agg
    for r in csv.DictReader(open(f[0])):
        agg[base(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    os.remove(f[0])
    return agg


what = sys.argv[1:] or ["stats", "traffic", "wait"]

if "stats" in what:
    with open(os.path.join(PROF, "r06_bench_default.json"), "w") as fo:
        d1 = prof("prof_stats", ["--kernel-trace", "--stats"], DRIVER, stdout=fo)
    for f in glob.glob(os.path.join(d1, "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(PROF, "r06_bench_default_kernel_stats.csv"))
    tr = glob.glob(os.path.join(d1, "**", "*kernel_trace.csv"), recursive=True)
    if tr:
        rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
        ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"])) for r in rows]
        counted = [i for i, k in enumerate(ks) if k[2].startswith("k_trace<false, true>")]
        if counted:
            lo, hi = counted[0], counted[-1]
            while lo > 0 and ks[lo][0] - ks[lo - 1][1] < 50e6:      # back to the host-side gap before this render (scene set-up, CPU baseline)
                lo -= 1
            while hi + 1 < len(ks) and ks[hi + 1][0] - ks[hi][1] < 50e6:
                hi += 1
            t0 = ks[lo][0]
            with open(os.path.join(PROF, "r06_instrumented_render_launches.csv"), "w") as fo:
                fo.write("start_ms,duration_us,kernel,grid_threads\n")
                for s, e, n, g in ks[lo:hi + 1]:
                    if not n.startswith("__amd_rocclr"):
                        fo.write("%.4f,%.2f,\"%s\",%d\n" % ((s - t0) / 1e6, (e - s) / 1e3, n, g))
            agg = collections.OrderedDict()
            for s, e, n, g in ks[lo:hi + 1]:
                # (k_tail is launched twice per batch since round 6: over the handed-over paths, 1024 workgroups, and — a few workgroups — over the
                # stragglers; the library's timer and bench.py's roofline keep the two apart, so does this summary)
                if n.startswith("k_tail<") and g < 1024 * 256:
                    n = n + " (stragglers)"
                a = agg.setdefault(n, [0, 0.0])
                a[0] += 1; a[1] += (e - s) / 1e6
            json.dump({"source": "rocprofv3 --kernel-trace of `python bench.py --steps 20 --warmup 5`: the launches between the host-side gaps around the render "
                                 "that launches k_trace<false, true> (= the instrumented render behind the roofline block)",
                       "span_ms": (ks[hi][1] - t0) / 1e6,
                       "kernels": {n: {"launches": c, "total_ms": round(t, 4), "avg_launch_ms": t / c} for n, (c, t) in agg.items() if not n.startswith("__amd_rocclr")}},
                      open(os.path.join(PROF, "r06_instrumented_render_kernels.json"), "w"), indent=1)
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
        if "<" in k["name"]:
            units[k["name"]] += k["units"]; launches[k["name"]] += k["launches"]   # (k_shade<common> / k_shade<rest> also on their own)
    rays = sum(s["rays"] for it in g.iterations for s in it["stats"])
    units["k_tail_rays"] = rays - units["k_trace"]                                            # rays traced AND shaded inside the tail
    # unit of the commit kernels = recorded vertex: the first iteration commits through k_commit, the rounds of the optimiser through
    # k_commit_records + k_splat_sorted (ppg_kernels.h "The commit of a ROUND")
    v_all = sum(s["vertices_committed"] for it in g.iterations for s in it["stats"])
    v_round = sum(s["vertices_committed"] for it in g.iterations for s in it["stats"] if it["iter"] > 0) if units.get("k_commit_records") else 0
    units["k_commit"] = v_all - v_round
    units["k_commit_records"] = units["k_splat_sorted"] = v_round
    del g, e

if "traffic" in what:
    a2 = counters(prof("pmc_fetch", ["--pmc", "FETCH_SIZE"], ONE))
    a3 = counters(prof("pmc_write", ["--pmc", "WRITE_SIZE", "TCC_HIT", "TCC_MISS"], ONE))
    KB = 1024.0
    gen_known_wr = 80.0 * units["k_generate"]                                           # ray_o, ray_d, thr, li, misc
    film_known_rd = 16.0 * W * H * SPP * STEPS + (4 + 11 * 4) * units["k_film"]          # every li sample once + per launch and pixel: index + 11 accumulators
    wr_factor = gen_known_wr / (a3["k_generate"]["WRITE_SIZE"] * KB) if a3["k_generate"]["WRITE_SIZE"] else None
    rd_factor = film_known_rd / (a2["k_film"]["FETCH_SIZE"] * KB) if a2["k_film"]["FETCH_SIZE"] else None
    raw_factors = {"k_film": rd_factor, "k_generate": wr_factor}
    if os.environ.get("PPG_PATH_LAYOUT", "") != "soa":
        # the automatic layout of this workload interleaves the path state: k_film / k_generate then touch 16 / 80 bytes of every 128-byte record and
        # no longer stream a known byte count — the factors measured with PPG_PATH_LAYOUT=soa in this round (2.053 / 1.000) are applied instead
        rd_factor, wr_factor = 2.0531788133671434, 1.0
    res = {"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE TCC_HIT TCC_MISS, one %d-pass render of the bench workload (kitchen-improved-720p)" % STEPS,
           "calibration": {"read_factor": rd_factor, "write_factor": wr_factor, "raw_factors_of_this_collection": raw_factors,
                           "note": "factor = known streamed bytes / (counter * 1024); MI355X_MICROARCH.md §HBM expects ~2 for reads"},
           "unit_of_work": {"k_trace": "traced ray", "k_shade<common>": "traced ray that hit one of the common material classes", "k_shade<rest>": "any other traced ray", "k_tail": "ray traced and shaded inside the tail (the same unit as the roofline's)",
                            "k_commit": "recorded vertex", "k_commit_records": "recorded vertex", "k_splat_sorted": "recorded vertex", "k_adam_apply": "record position of the round (holes included)"},
           "bytes_per_unit": {}, "per_kernel": {}}
    for k in sorted(set(a2) | set(a3)):
        rd = a2[k].get("FETCH_SIZE", 0.0) * KB * (rd_factor or 2.0)
        wr = a3[k].get("WRITE_SIZE", 0.0) * KB * (wr_factor or 1.0)
        hit, miss = a3[k].get("TCC_HIT", 0.0), a3[k].get("TCC_MISS", 0.0)
        u = units.get("k_tail_rays") if k == "k_tail" else units.get(k)
        res["per_kernel"][k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "l2_hit_rate": hit / (hit + miss) if hit + miss else None, "units": u, "launches": launches.get(k)}
        if u:
            res["bytes_per_unit"][k] = (rd + wr) / u
    json.dump(res, open(os.path.join(PROF, "r06_pmc_traffic_kitchen.json"), "w"), indent=1)
    print(json.dumps(res["bytes_per_unit"], indent=1))

if "wait" in what:
    SQ = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU"]
    a = counters(prof("pmc_sq", ["--pmc"] + SQ, ONE))
    out = {}
    for k in ("k_trace", "k_shade<common>", "k_shade<rest>", "k_tail", "k_commit", "k_commit_records", "k_splat_sorted", "k_sort_slices", "k_adam_apply"):
        c = a.get(k)
        if not c:
            continue
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        u = units.get("k_tail_rays") if k == "k_tail" else units.get(k)
        out[k] = dict(c, wait_any_share=c.get("SQ_WAIT_ANY", 0.0) / wc, wait_inst_share=c.get("SQ_WAIT_INST_ANY", 0.0) / wc, active_share=c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc,
                      units=u, valu_wave_instructions_per_unit=(c.get("SQ_INSTS_VALU", 0.0) / u) if u else None)
    json.dump({"source": "rocprofv3 --pmc " + " ".join(SQ) + ", one %d-pass render of kitchen-improved-720p; quad-cycle units" % STEPS, "kernels": out},
              open(os.path.join(PROF, "r06_pmc_wait_cycles.json"), "w"), indent=1)
    print(json.dumps({k: {m: round(v[m], 3) for m in ("wait_any_share", "wait_inst_share", "active_share")} for k, v in out.items()}, indent=1))
