"""One-off GPU-box check: bit-exact parity against the oracle at a larger size than the test suite uses, over the feature combinations
(full material set, null surfaces, environment emitter, next-event estimation, improved preset, unbounded depth)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
import ppg_host
from conftest import CBOX_PROPS, IMPROVED, ORACLE_SO, make_oracle
import test_gpu_parity as T

lib = ctypes.CDLL(ORACLE_SO)
W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "640x360").split("x"))
cases = []
s1 = T._full_materials_scene((W, H)); cases.append(("materials/improved/kickstart", s1, dict(IMPROVED, nee="kickstart", maxDepth=14, rrDepth=5)))
s2 = T._pane_scene((W, H)); s2.environment = (0.3, 0.4, 0.6); cases.append(("panes+env/always/unbounded", s2, dict(nee="always", maxDepth=-1, rrDepth=4, strictNormals=0)))
s3 = T._pane_scene((W, H)); s3.materials[-1] = dict(type="diffuse", reflectance=(0.7, 0.7, 0.7), twosided=True, opacity=(0.35, 0.4, 0.45))
cases.append(("mask/box-box/var", s3, dict(spatialFilter="box", directionalFilter="box", bsdfSamplingFractionLoss="var", sTreeThreshold=2000, nee="kickstart")))
s4 = ppg_host.room_scene(W, H, n_boxes=400, tess=4, glossy=True); cases.append(("room-glossy/improved", s4, dict(IMPROVED, maxDepth=12, rrDepth=5, strictNormals=0)))
ok = True
for name, scene, extra in cases:
    props = dict(CBOX_PROPS, budget=28, seed=7); props.update(extra)
    if "sppPerPass" in extra: props["budget"] = 31
    g, o = ppg_host.Engine.hip(**props), make_oracle(lib, threads=os.cpu_count() or 8, **props)
    t0 = time.time(); ig = ppg_host.GuidedPathTracer(engine=g).render(scene); tg = time.time() - t0
    t0 = time.time(); io = ppg_host.GuidedPathTracer(engine=o).render(scene); to = time.time() - t0
    same = np.array_equal(ig, io, equal_nan=True)
    a, b = g.read_sdtree(), o.read_sdtree()
    tree = np.array_equal(a["children"], b["children"]) and np.array_equal(a["sampling"]["node_sums"], b["sampling"]["node_sums"]) and np.array_equal(a["theta"], b["theta"])
    print("%-32s image %s tree %s  gpu %.2fs cpu %.1fs  mean %.4f leaves %d" % (name, same, tree, tg, to, float(np.nanmean(ig)), a["n_leaves"]), flush=True)
    ok &= same and tree
print("ALL EQUAL" if ok else "MISMATCH")
