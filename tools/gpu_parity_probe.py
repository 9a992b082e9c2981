#!/usr/bin/env python3
"""Debug helper (GPU box): step CBOX through both engines and print where they first differ."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd"))
import numpy as np
import torch  # noqa: F401  (its HIP runtime must be loaded before libppg_hip.so)
import ppg_host

res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
passes = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4]
extra = {}
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    extra[k] = int(v) if v.lstrip("-").isdigit() else v
props = dict(budgetType="spp", budget=4 * sum(passes), maxDepth=10, rrDepth=10, strictNormals=1, seed=42)
props.update(extra)
scene = ppg_host.cbox_scene(res, res)
g = ppg_host.Engine.hip(**props); g.set_scene(scene)
o = ppg_host.Engine(os.path.join(ROOT, "oracle", "libppg_oracle.so"), "ppgo_", **props); o.set_scene(scene)
o.lib.ppgo_set_modes(o.ctx, 0, 0, 8)
g.begin_render(); o.begin_render()
for it, p in enumerate(passes):
    fin = it == len(passes) - 1
    g.begin_iteration(fin); o.begin_iteration(fin)
    tb_g, tb_o = g.read_sdtree(), o.read_sdtree()
    print("iter", it, "reset: stree equal", np.array_equal(tb_g["children"], tb_o["children"]),
          "building topo equal", np.array_equal(tb_g["building"]["node_children"], tb_o["building"]["node_children"]),
          "nodes", tb_g["building"]["node_children"].shape, tb_o["building"]["node_children"].shape)
    t0 = time.time(); sg = g.render_passes(p); tg = time.time() - t0
    t0 = time.time(); so = o.render_passes(p); to = time.time() - t0
    print("  gpu ", sg.as_dict(), "%.3fs" % tg)
    print("  orac", so.as_dict(), "%.3fs" % to)
    tb_g, tb_o = g.read_sdtree(), o.read_sdtree()
    fx_g, fx_o = tb_g["building"]["node_fixed"], tb_o["building"]["node_fixed"]
    print("  building acc equal", fx_g.shape == fx_o.shape and np.array_equal(fx_g, fx_o),
          "weights equal", np.array_equal(tb_g["building"]["stat_weight"], tb_o["building"]["stat_weight"]))
    if fx_g.shape == fx_o.shape and not np.array_equal(fx_g, fx_o):
        d = np.argwhere(fx_g != fx_o)
        print("   first diffs", d[:5], fx_g[tuple(d[0])], fx_o[tuple(d[0])], "n diff", len(d), "of", fx_g.size)
    tg_, to_ = g.build_sdtree(), o.build_sdtree()
    print("  tree gpu ", tg_.as_dict()); print("  tree orac", to_.as_dict())
    g.end_iteration(); o.end_iteration()
    fg, fo = g.read_film(), o.read_film()
    print("  film equal", np.array_equal(fg, fo), "max abs diff", np.abs(fg - fo).max(), "mean", fg.mean(), fo.mean(), "theta eq", np.array_equal(g.read_sdtree()["theta"], o.read_sdtree()["theta"]))
