#!/usr/bin/env python3
"""GPU box: rocprofv3 kernel stats + HBM traffic (PMC) of the bench workload → profiles/r01_*.

Runs, each as its own rocprofv3 invocation (counters never share a run with tracing, and FETCH_SIZE /
WRITE_SIZE need separate passes: MI355X_MICROARCH.md §rocprofv3 PMC slots):
  1. --kernel-trace --stats          → profiles/r01_kernel_stats.csv
  2. --pmc FETCH_SIZE                → bytes read through the L2's memory side, per kernel
  3. --pmc WRITE_SIZE TCC_HIT TCC_MISS
Calibration (the guide: FETCH_SIZE under-reports wide streaming reads by 2x on gfx950, WRITE_SIZE is
uncalibrated): k_generate writes exactly 80 B per path and k_film reads exactly 16 B per sample + 48 B per pixel and launch, both
pure streams; their known byte counts give the read / write correction factors applied to all kernels.
Output: profiles/r01_pmc_traffic.json {bytes_per_unit: {kernel: HBM bytes per unit}, ...}.
"""
import collections, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 31
W, H, SPP = 1280, 720, 4
BENCH = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(STEPS), "--warmup", "0", "--no-cpu", "--no-roofline"]
env = dict(os.environ, TMPDIR="/tmp")


def prof(tag, args):
    d = os.path.join(OUT, tag)
    subprocess.run(["rocprofv3"] + args + ["--output-format", "csv", "-d", d, "-o", "p", "--"] + BENCH, cwd="/tmp", env=env,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    return d


def counters(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    f = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not f:
        return agg, calls
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), r["Counter_Name"])
        if key not in seen and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            seen.add(key); calls[k] += 1
    os.remove(f[0])
    return agg, calls


d1 = prof("prof_stats", ["--kernel-trace", "--stats"])
for f in glob.glob(os.path.join(d1, "*kernel_trace.csv")):
    os.remove(f)
a2, c2 = counters(prof("pmc_fetch", ["--pmc", "FETCH_SIZE"]))
a3, c3 = counters(prof("pmc_write", ["--pmc", "WRITE_SIZE", "TCC_HIT", "TCC_MISS"]))

# units per kernel over the run (from one un-profiled run with kernel timing)
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))
import torch  # noqa: F401,E402
import ppg_host  # noqa: E402
e = ppg_host.Engine.hip(budgetType="spp", sppPerPass=SPP, maxDepth=10, rrDepth=10, strictNormals=1, seed=1234, budget=float(STEPS * SPP))
e.set_scene(ppg_host.cbox_scene(W, H)); e.enable_kernel_timing(True)
g = ppg_host.GuidedPathTracer(engine=e); g.render()
units = {k["name"].split("<")[0]: k["units"] for k in e.kernel_times()}
launches = {k["name"].split("<")[0]: k["launches"] for k in e.kernel_times()}
committed = sum(s["vertices_committed"] for it in g.iterations for s in it["stats"])
units["k_commit"] = committed  # unit of k_commit = recorded vertex

KB = 1024.0
gen_known_wr = 80.0 * units["k_generate"]                       # ray_o, ray_d, thr, li, misc
film_known_rd = 16.0 * W * H * SPP * STEPS + (4 + 11 * 4) * units["k_film"]  # every li sample once + per launch and pixel: index + 11 accumulators
wr_factor = gen_known_wr / (a3["k_generate"]["WRITE_SIZE"] * KB) if a3["k_generate"]["WRITE_SIZE"] else None
rd_factor = film_known_rd / (a2["k_film"]["FETCH_SIZE"] * KB) if a2["k_film"]["FETCH_SIZE"] else None
res = {"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE TCC_HIT TCC_MISS, bench.py --steps %d (cbox-720p)" % STEPS,
       "calibration": {"read_factor_from_k_film": rd_factor, "write_factor_from_k_generate": wr_factor,
                       "note": "factor = known streamed bytes / (counter * 1024); MI355X_MICROARCH.md §HBM expects ~2 for reads"},
       "bytes_per_unit": {}, "per_kernel": {}}
for k in sorted(set(a2) | set(a3)):
    rd = a2[k].get("FETCH_SIZE", 0.0) * KB * (rd_factor or 2.0)
    wr = a3[k].get("WRITE_SIZE", 0.0) * KB * (wr_factor or 1.0)
    hit, miss = a3[k].get("TCC_HIT", 0.0), a3[k].get("TCC_MISS", 0.0)
    u = units.get(k)
    res["per_kernel"][k] = {"hbm_read_bytes": rd, "hbm_write_bytes": wr, "l2_hit_rate": hit / (hit + miss) if hit + miss else None,
                            "units": u, "launches": launches.get(k)}
    if u:
        res["bytes_per_unit"][k] = (rd + wr) / u
os.makedirs(os.path.join(ROOT, "gpurun_out", "profiles"), exist_ok=True)
json.dump(res, open(os.path.join(OUT, "profiles", "r01_pmc_traffic.json"), "w"), indent=1)
for f in glob.glob(os.path.join(d1, "*kernel_stats.csv")):
    os.replace(f, os.path.join(OUT, "profiles", "r01_kernel_stats.csv"))
print(json.dumps(res, indent=1)[:3000])
