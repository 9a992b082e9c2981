"""Wall time per phase of render() (begin_iteration = S-tree refine on the host mirror + upload + D-tree reset;
passes; finish = variance; build).  usage: phase_timing.py [cbox|room|kitchen] [passes] [world]
world > 1: time rank 0's share of a `world`-way tile shard (no collectives: the compute + fixed host part of one rank)."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'practical-path-guiding_amd'))
sys.path.insert(0, ROOT)
import torch, ppg_host  # noqa
which = sys.argv[1] if len(sys.argv) > 1 else "cbox"
n_pass = int(sys.argv[2]) if len(sys.argv) > 2 else (255 if which == "cbox" else 127)
WORLD = int(sys.argv[3]) if len(sys.argv) > 3 else 1
if which == "cbox":
    props = dict(budgetType="spp", sppPerPass=4, maxDepth=10, rrDepth=10, strictNormals=1, seed=1234)
    scene = ppg_host.cbox_scene(1280, 720)
elif which == "kitchen":
    from bench import KITCHEN_FILE, scene_props
    scene = ppg_host.load_scene_file(KITCHEN_FILE)
    props = scene_props(KITCHEN_FILE, dict(budgetType="spp", seed=1234))
else:
    props = dict(budgetType="spp", sppPerPass=1, seed=1234, sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic",
                 directionalFilter="box", sTreeThreshold=4000)
    scene = ppg_host.room_scene(1280, 720, n_boxes=1820, tess=8)
props["budget"] = float(n_pass * props["sppPerPass"])
passes, it, left = [], 0, n_pass
while left > 0:
    p = min(left, 1 << it)
    if left - p < 2 * p:
        p = left
    passes.append(p); left -= p; it += 1
for rep in range(2):
    e = ppg_host.Engine.hip(**props); e.set_scene(scene)
    if WORLD > 1:
        e.set_shard(0, WORLD, 32)
    t, per_it = {}, []
    def T(name, f, *a):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(*a); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        t[name] = t.get(name, 0) + dt; return dt
    t0 = time.perf_counter()
    T('begin_render', e.begin_render)
    for it, p in enumerate(passes):
        a = T('begin_iteration', e.begin_iteration, it == len(passes) - 1)
        b = T('passes', e.render_passes_nostat, p)
        T('finish', e.finish_passes)
        c = T('build', e.build_sdtree)
        T('end_it', e.end_iteration)
        per_it.append((p, round(a * 1e3, 2), round(b * 1e3, 2), round(c * 1e3, 2), e.sdtree_info().n_leaves))
    T('end_render', e.end_render)
    tot = time.perf_counter() - t0
    print(rep, which, 'world', WORLD, 'total %.1f ms' % (tot * 1e3), {k: round(v * 1e3, 2) for k, v in t.items()}, 'Msamples/s %.1f' % (scene.camera["width"] * scene.camera["height"] * props["budget"] / tot / 1e6))
print("per iteration (passes, begin_iteration ms, passes ms, build ms, leaves):", per_it)
