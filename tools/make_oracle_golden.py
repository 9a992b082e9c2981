#!/usr/bin/env python3
"""Writes tests/golden/oracle_cbox_*.npz: small renders of the procedural CBOX by the CPU oracle
(film, SD-tree topology and sums, per-iteration statistics).  The GPU parity tests compare the HIP path
against these committed vectors bit for bit, and tests/test_host_logic.py checks that the oracle still
reproduces them.  Re-run after any deliberate change of the numerical contract."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))
import numpy as np
import ppg_host

lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libppg_oracle.so"))
BASE = dict(budgetType="spp", maxDepth=10, rrDepth=10, strictNormals=1)
CASES = {
    "default": (dict(), 48, 60, 7),
    "improved": (dict(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box",
                      sTreeThreshold=4000, sppPerPass=1), 48, 63, 11),
    "boxbox": (dict(spatialFilter="box", directionalFilter="box", bsdfSamplingFractionLoss="var", sTreeThreshold=600, sampleCombination="discard"), 32, 28, 13),
}
for name, (extra, res, budget, seed) in CASES.items():
    e = ppg_host.Engine(lib, "ppgo_", budget=float(budget), seed=seed, **dict(BASE, **extra))
    lib.ppgo_set_modes(e.ctx, 0, 0, 8)
    gpt = ppg_host.GuidedPathTracer(engine=e)
    film = gpt.render(ppg_host.cbox_scene(res, res))
    t = e.read_sdtree()
    stats = np.array([[s["rays"], s["path_length_sum"], s["vertices_committed"]] for it in gpt.iterations for s in it["stats"]], np.uint64)
    var = np.array([s["variance"] for it in gpt.iterations for s in it["stats"]], np.float32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_cbox_%s.npz" % name), film=film, res=res, budget=budget, seed=seed,
                        stree_children=t["children"], stree_axis=t["axis"], dtree_children=t["sampling"]["node_children"],
                        dtree_sums=t["sampling"]["node_sums"], dtree_num=t["sampling"]["num_nodes"], theta=t["theta"], stats=stats, variance=var,
                        passes=np.array([it["passes"] for it in gpt.iterations]))
    print(name, film.mean(), t["children"].shape, t["sampling"]["node_children"].shape, var)
