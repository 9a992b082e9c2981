"""Host logic on CPU: the Python mirror of render() equals the single-call render(), oracle determinism
across thread counts (fixed-point accumulation), sharded rendering over gloo equals single-rank rendering,
.sdt dump format, and the committed golden vectors still match the oracle."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from conftest import CBOX_PROPS, GOLDEN, IMPROVED, ROOT, make_oracle


def _tree_equal(a, b):
    if not (np.array_equal(a["children"], b["children"]) and np.array_equal(a["axis"], b["axis"])):
        return False
    for k in ("sampling", "building"):
        for f in ("num_nodes", "node_children", "node_sums", "max_depth"):
            if not np.array_equal(a[k][f], b[k][f]):
                return False
    return np.array_equal(a["theta"], b["theta"])


def test_python_driver_equals_single_call_render(oracle_lib):
    import ppg_host
    scene = ppg_host.cbox_scene(48, 48)
    for extra in (dict(), IMPROVED):
        props = dict(CBOX_PROPS, budget=60, seed=9, **extra)
        a = make_oracle(oracle_lib, **props); a.set_scene(scene); a.render(); img_a = a.read_film()
        b = make_oracle(oracle_lib, **props)
        img_b = ppg_host.GuidedPathTracer(engine=b).render(scene)
        assert np.array_equal(img_a, img_b)
        assert _tree_equal(a.read_sdtree(), b.read_sdtree())


def test_fixed_point_accumulation_is_thread_count_invariant(oracle_lib):
    import ppg_host
    scene = ppg_host.cbox_scene(40, 40)
    outs = []
    for threads in (1, 7):
        e = make_oracle(oracle_lib, threads=threads, budget=60, seed=3, **dict(CBOX_PROPS, **IMPROVED))
        e.set_scene(scene); e.render()
        outs.append((e.read_film(), e.read_sdtree()))
    assert np.array_equal(outs[0][0], outs[1][0]) and _tree_equal(outs[0][1], outs[1][1])


def test_float_accumulation_mode_is_statistically_the_same(oracle_lib):
    # acc_mode FLOAT = the reference's sequential float adds (GP:59-62); FIXED must not change the estimate
    import ppg_host
    scene = ppg_host.cbox_scene(48, 48)
    imgs = []
    for acc in (0, 1):
        e = make_oracle(oracle_lib, threads=1, acc=acc, budget=28, seed=5, **CBOX_PROPS)
        e.set_scene(scene); e.begin_render()
        e.begin_iteration(False); e.render_passes(1); t = e.build_sdtree(); e.end_iteration()
        imgs.append((e.read_film(), t.avg_stat_weight, t.avg_mean_radiance))
    assert np.array_equal(imgs[0][0], imgs[1][0])  # iteration 0 does not depend on the SD-tree
    assert imgs[0][1] == imgs[1][1] and abs(imgs[0][2] / imgs[1][2] - 1) < 1e-4


def test_sdt_dump_format(oracle_lib, tmp_path):
    # byte layout of guided_path.cpp:1197-1205 + 699-711, as read by visualizer/src/main.cpp:142-176
    import ppg_host
    e = make_oracle(oracle_lib, budget=12, seed=1, **CBOX_PROPS)
    e.set_scene(ppg_host.cbox_scene(32, 32)); e.render()
    path = str(tmp_path / "t.sdt")
    e.dump_sdtree(path)
    buf = open(path, "rb").read()
    cam = struct.unpack_from("<16f", buf, 0)
    assert abs(cam[3] - 278) < 1e-3 and abs(cam[11] + 800) < 1e-3  # camera-to-world translation column
    off, trees = 64, 0
    tree = e.read_sdtree()
    while off < len(buf):
        px, py, pz, sx, sy, sz, mean = struct.unpack_from("<7f", buf, off); off += 28
        sw, nn = struct.unpack_from("<QQ", buf, off); off += 16
        assert sx > 0 and sw > 0 and 1 <= nn <= 65536 and mean >= 0
        for _ in range(nn * 4):
            s, c = struct.unpack_from("<fH", buf, off); off += 6
            assert s >= 0 and c < nn
        trees += 1
    assert off == len(buf) and trees == int((tree["sampling"]["stat_weight"] > 0).sum())


def test_golden_vectors_match_oracle(oracle_lib):
    # tests/golden/oracle_cbox_*.npz were written by tools/make_oracle_golden.py; the GPU tests compare against the same files
    import ppg_host
    g = np.load(os.path.join(GOLDEN, "oracle_cbox_default.npz"))
    e = make_oracle(oracle_lib, budget=float(g["budget"]), seed=int(g["seed"]), **CBOX_PROPS)
    e.set_scene(ppg_host.cbox_scene(int(g["res"]), int(g["res"]))); e.render()
    assert np.array_equal(e.read_film(), g["film"])
    t = e.read_sdtree()
    assert np.array_equal(t["children"], g["stree_children"]) and np.array_equal(t["sampling"]["node_children"], g["dtree_children"])
    assert np.array_equal(t["sampling"]["node_sums"], g["dtree_sums"])


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "..", "tests"))
import ctypes, numpy as np, torch.distributed as dist
import ppg_host
from ppg_host.distributed import HostReducer
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = ctypes.CDLL(sys.argv[2])
props = dict(budgetType="spp", maxDepth=10, rrDepth=10, strictNormals=1, budget=60, seed=17)
if sys.argv[4] == "inversevar":
    props.update(sampleCombination="inversevar", spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=2000)
if sys.argv[4] == "improved":
    props.update(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=2000, sppPerPass=1)
if sys.argv[4] in ("nee", "full-scene"):
    props.update(nee="kickstart", bsdfSamplingFractionLoss="var", spatialFilter="box", sTreeThreshold=2000)
e = ppg_host.Engine(lib, "ppgo_", **props)
lib.ppgo_set_modes(e.ctx, 0, 0, 2)
scene = ppg_host.cbox_scene(64, 48)
if sys.argv[4] == "full-scene":
    from test_gpu_parity import _sphere_scene
    scene = _sphere_scene((64, 48), sky=True)
e.set_scene(scene); e.set_shard(rank, world, 16)
img = ppg_host.GuidedPathTracer(engine=e, reducer=HostReducer(dist, gather_all=(sys.argv[5] == "gather"))).render()
t = e.read_sdtree()
np.savez(os.path.join(sys.argv[3], "rank%d.npz" % rank), film=img, children=t["children"], dch=t["sampling"]["node_children"], dsum=t["sampling"]["node_sums"], theta=t["theta"])
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode,world,scheme", [("default", 2, "owner"), ("inversevar", 2, "owner"), ("improved", 2, "owner"), ("nee", 2, "owner"),
                                               ("full-scene", 2, "owner"), ("improved", 4, "owner"), ("nee", 4, "owner"), ("improved", 2, "gather")])
def test_sharded_render_over_gloo_equals_single_rank(oracle_lib, tmp_path, mode, world, scheme):
    """world_size 2 and 4, gloo: tiles sharded, SD-tree statistics all-reduced as int64, the optimiser's records sent to the OWNER of
    their D-tree (all-to-all) and the owners' optimiser state all-gathered ("owner"; "gather": round 2's gather-everything scheme) →
    the merged render, the SD-tree and the learned fractions are bit-identical to the unsharded ones on every rank."""
    import ppg_host
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], str(script), os.path.join(ROOT, "practical-path-guiding_amd"),
           os.path.join(ROOT, "oracle", "libppg_oracle.so"), str(tmp_path), mode, scheme]
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    props = dict(CBOX_PROPS, budget=60, seed=17)
    if mode == "inversevar":
        props.update(sampleCombination="inversevar", spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=2000)
    if mode == "improved":  # learned BSDF sampling fraction: the per-pass Adam sums are all-reduced through the pass hook
        props.update(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box",
                     sTreeThreshold=2000, sppPerPass=1)
    if mode in ("nee", "full-scene"):  # direct-light vertices are committed inside Li's loop on whichever rank owns the pixel
        props.update(nee="kickstart", bsdfSamplingFractionLoss="var", spatialFilter="box", sTreeThreshold=2000)
    scene = ppg_host.cbox_scene(64, 48)
    if mode == "full-scene":  # analytic spheres (glass, rough gold, lamp) under an emitting sky dome: the S-tree spans the dome
        from test_gpu_parity import _sphere_scene
        scene = _sphere_scene((64, 48), sky=True)
    e = make_oracle(oracle_lib, threads=4, **props)
    e.set_scene(scene); e.render()
    ref_img, ref_t = e.read_film(), e.read_sdtree()
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(got["children"], ref_t["children"])
        assert np.array_equal(got["dch"], ref_t["sampling"]["node_children"]) and np.array_equal(got["dsum"], ref_t["sampling"]["node_sums"])
        assert np.array_equal(got["film"], ref_img)
        assert np.array_equal(got["theta"], ref_t["theta"])


def _time_budget_checks(make_engine, scene, budget):
    """renderTime (GP:1434-1514): iterations of 1, 2, 4, ... passes until the budget is spent; with sampleCombination = automatic the
    last iteration keeps rendering batches of its own size until the time is up (GP:1482-1501)."""
    import time
    import ppg_host
    for combo in ("automatic", "discard"):
        e = make_engine(budgetType="seconds", budget=budget, sampleCombination=combo, seed=5, maxDepth=6)
        g = ppg_host.GuidedPathTracer(engine=e)
        t0 = time.monotonic()
        img = g.render(scene)
        dt = time.monotonic() - t0
        passes = [it["passes"] for it in g.iterations]
        assert passes == [1 << k for k in range(len(passes))] and len(passes) >= 2
        assert dt >= budget * 0.98 and np.isfinite(img).all() and img.mean() > 0.01
        # inside an iteration the reference aborts only once the elapsed time in WHOLE seconds exceeds the budget (GP:1259-1262)
        done = g.iterations[-1]["stats"][-1]["passes_rendered_total"]
        assert sum(passes[:-1]) < done
        if combo == "discard":
            assert done <= sum(passes) and all(len(it["stats"]) == 1 for it in g.iterations)
        assert dt < budget + 1.5
        # the single-call entry point runs the same loop
        e2 = make_engine(budgetType="seconds", budget=budget / 2, sampleCombination=combo, seed=5, maxDepth=6)
        e2.set_scene(scene)
        t0 = time.monotonic(); e2.render(); dt2 = time.monotonic() - t0
        assert dt2 >= budget / 2 * 0.98 and np.isfinite(e2.read_film()).all()


def test_time_budget_on_the_oracle(oracle_lib):
    import ppg_host
    _time_budget_checks(lambda **p: make_oracle(oracle_lib, threads=4, **p), ppg_host.cbox_scene(24, 24), 0.6)


def test_automatic_sdt_dumps_and_memory_cap(oracle_lib, tmp_path):
    import ppg_host
    scene = ppg_host.cbox_scene(32, 32)
    prefix = str(tmp_path / "run")
    e = make_oracle(oracle_lib, budget=60, seed=2, dumpSDTree=1, dumpPrefix=prefix, **CBOX_PROPS)
    g = ppg_host.GuidedPathTracer(engine=e); g.render(scene)
    # "<dest>-%02d.sdt" after every training iteration, none for the final one (GP:1191-1195, 1417-1420)
    files = sorted(os.listdir(tmp_path))
    assert files == ["run-%02d.sdt" % k for k in range(len(g.iterations) - 1)]
    assert all(os.path.getsize(tmp_path / f) > 64 for f in files)
    # sdTreeMaxMemory (GP:957-962): once the footprint estimate reaches the cap the S-tree stops being refined
    free = make_oracle(oracle_lib, budget=124, seed=2, sTreeThreshold=200, **CBOX_PROPS); free.set_scene(scene); free.render()
    capped = make_oracle(oracle_lib, budget=124, seed=2, sTreeThreshold=200, sdTreeMaxMemory=0, **CBOX_PROPS); capped.set_scene(scene); capped.render()
    assert capped.read_sdtree()["n_leaves"] == 1 < free.read_sdtree()["n_leaves"]  # footprint / 1e6 >= 0 always: never refined


def test_torus_scene_is_closed_and_deterministic():
    """The torus-class stand-in: triangle count by formula, every edge of the torus and of the cube shared by exactly two triangles
    (closed surfaces — the glass cube must be watertight for the dielectric's inside / outside bookkeeping), one emitter."""
    import ppg_host
    a, b = ppg_host.torus_scene(64, 36, n_major=24, n_minor=12), ppg_host.torus_scene(64, 36, n_major=24, n_minor=12)
    assert a.n_triangles == 2 * 24 * 12 + 12 + 12 + 2 and np.array_equal(a.positions, b.positions) and np.array_equal(a.indices, b.indices)
    assert (a.tri_emitter >= 0).sum() == 2 and len(a.emitters) == 1
    for mat in (1, 2):                                     # glass cube, torus
        tris = a.indices[a.tri_material == mat]
        pos = np.round(a.positions[tris], 5)
        edges = {}
        for t in pos:
            for k in range(3):
                e = tuple(sorted([tuple(t[k]), tuple(t[(k + 1) % 3])]))
                edges[e] = edges.get(e, 0) + 1
        assert set(edges.values()) == {2}, mat
