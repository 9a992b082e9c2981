"""Host logic on CPU: the Python mirror of render() equals the single-call render(), oracle determinism
across thread counts (fixed-point accumulation), sharded rendering over gloo equals single-rank rendering,
.sdt dump format, and the committed golden vectors still match the oracle."""
import ctypes
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


@pytest.mark.parametrize("case,extra", [
    ("default", {}),
    ("improved", dict(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=4000, sppPerPass=1)),
    ("boxbox", dict(spatialFilter="box", directionalFilter="box", bsdfSamplingFractionLoss="var", sTreeThreshold=600, sampleCombination="discard"))])
def test_golden_vectors_match_oracle(oracle_lib, case, extra):
    # tests/golden/oracle_cbox_*.npz were written by tools/make_oracle_golden.py; the GPU tests compare against the same files
    # ("improved": a final iteration of 32 passes = two groups of 16, include/ppg.h "Final iteration: groups of passes")
    import ppg_host
    g = np.load(os.path.join(GOLDEN, "oracle_cbox_%s.npz" % case))
    e = make_oracle(oracle_lib, threads=8, budget=float(g["budget"]), seed=int(g["seed"]), **dict(CBOX_PROPS, **extra))
    e.set_scene(ppg_host.cbox_scene(int(g["res"]), int(g["res"]))); e.render()
    assert np.array_equal(e.read_film(), g["film"], equal_nan=True)
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
if sys.argv[4] == "auto-final":
    props.update(budget=2044, seed=2, sampleCombination="automatic")
if sys.argv[4] in ("stragglers", "regions"):
    props.update(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=2000, sppPerPass=1,
                 maxDepth=-1, rrDepth=3, budget=63)
e = ppg_host.Engine(lib, "ppgo_", **props)
if sys.argv[4] == "stragglers":
    e._call("debug_set_defer_depth", ctypes.c_int32(6))
lib.ppgo_set_modes(e.ctx, 0, 0, 2)
if sys.argv[4] == "regions":
    e.set_adam_regions(5)
scene = ppg_host.cbox_scene(24, 16) if sys.argv[4] == "auto-final" else ppg_host.cbox_scene(64, 48)
if sys.argv[4] == "full-scene":
    from test_gpu_parity import _sphere_scene
    scene = _sphere_scene((64, 48), sky=True)
e.set_scene(scene); e.set_shard(rank, world, 8 if sys.argv[4] == "auto-final" else 16)
gpt = ppg_host.GuidedPathTracer(engine=e, reducer=HostReducer(dist, gather_all=(sys.argv[5] == "gather")))
img = gpt.render()
if sys.argv[4] == "auto-final":
    assert gpt.iterations[-1].get("final_passes") == 256, [(it["passes"], it.get("final_passes")) for it in gpt.iterations]
    own = gpt.iterations[-1]["stats"][-1]["samples"]  # 16 groups of 16 passes over the whole film: every rank rendered 16 / world of them
    assert own == 24 * 16 * 4 * 256 // world, own
t = e.read_sdtree()
np.savez(os.path.join(sys.argv[3], "rank%d.npz" % rank), film=img, children=t["children"], dch=t["sampling"]["node_children"], dsum=t["sampling"]["node_sums"], theta=t["theta"])
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode,world,scheme", [("default", 2, "owner"), ("inversevar", 2, "owner"), ("improved", 2, "owner"), ("nee", 2, "owner"),
                                               ("full-scene", 2, "owner"), ("improved", 4, "owner"), ("nee", 4, "owner"), ("improved", 2, "gather"),
                                               ("auto-final", 2, "owner"), ("auto-final", 4, "owner"), ("stragglers", 2, "owner"), ("stragglers", 3, "owner"),
                                               ("stragglers", 2, "gather"), ("regions", 2, "owner"), ("regions", 3, "owner")])
def test_sharded_render_over_gloo_equals_single_rank(oracle_lib, tmp_path, mode, world, scheme):
    """world_size 2 and 4, gloo: tiles sharded, SD-tree statistics all-reduced as int64, the optimiser's records sent to the OWNER of
    their D-tree (all-to-all) and the owners' optimiser state all-gathered ("owner"; "gather": round 2's gather-everything scheme) →
    the merged render, the SD-tree and the learned fractions are bit-identical to the unsharded ones on every rank.
    The FINAL iteration is sharded by whole groups of passes, not by tiles (include/ppg.h "Final iteration: groups of passes"): "improved" has
    three groups (16 + 16 + 13 passes), "auto-final" sixteen — after the tile-sharded training passes of the same iteration, whose film the
    groups are added to (sampleCombination = automatic switching to FINAL in the middle of iteration 7, GP:1400-1411).
    "regions": rounds by image region (ppg_set_adam_regions) — R hook calls per pass on every rank, also on a rank that owns no tile of a group.
    "stragglers": unbounded paths, and the tests' switch lowers the depth of include/ppg.h "STRAGGLERS" to 6 — a quarter of the paths' records
    are applied one round late, those of an iteration's last round in a round of their own for which every rank's hook is called."""
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
    if mode == "auto-final":
        props.update(budget=2044, seed=2, sampleCombination="automatic")
        scene = ppg_host.cbox_scene(24, 16)
    if mode == "full-scene":  # analytic spheres (glass, rough gold, lamp) under an emitting sky dome: the S-tree spans the dome
        from test_gpu_parity import _sphere_scene
        scene = _sphere_scene((64, 48), sky=True)
    if mode in ("stragglers", "regions"):
        props.update(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=2000,
                     sppPerPass=1, maxDepth=-1, rrDepth=3, budget=63)
    e = make_oracle(oracle_lib, threads=4, **props)
    if mode == "regions":  # rounds by image region (include/ppg.h): five groups over 2 x 2 blocks -> clamped to four; a rank's tiles of a group may be none
        e.set_adam_regions(5)
    if mode == "stragglers":
        e._call("debug_set_defer_depth", ctypes.c_int32(6))
    e.set_scene(scene); e.render()
    if mode == "stragglers":
        hist = (ctypes.c_uint64 * 4096)()
        oracle_lib.ppgo_path_length_histogram(e.ctx, hist)
        assert sum(hist[7:]) > 10000  # (the test needs stragglers)
    ref_img, ref_t = e.read_film(), e.read_sdtree()
    for r in range(world):
        got = np.load(tmp_path / ("rank%d.npz" % r))
        assert np.array_equal(got["children"], ref_t["children"])
        assert np.array_equal(got["dch"], ref_t["sampling"]["node_children"]) and np.array_equal(got["dsum"], ref_t["sampling"]["node_sums"])
        assert np.array_equal(got["film"], ref_img)
        assert np.array_equal(got["theta"], ref_t["theta"])


CANCEL_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import ctypes, torch.distributed as dist
import ppg_host
from ppg_host.distributed import HostReducer, RenderAborted
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = ctypes.CDLL(sys.argv[2])
props = dict(budgetType="spp", maxDepth=10, rrDepth=10, strictNormals=1, budget=60, seed=17, sampleCombination="inversevar", bsdfSamplingFractionLoss="kl",
             spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=2000, sppPerPass=1)
e = ppg_host.Engine(lib, "ppgo_", **props)
lib.ppgo_set_modes(e.ctx, 0, 0, 2)
e.set_scene(ppg_host.cbox_scene(48, 32)); e.set_shard(rank, world, 16)
red = HostReducer(dist)
gpt = ppg_host.GuidedPathTracer(engine=e, reducer=red)
# rank `who` is cancelled after iteration 2; iteration 3 (8 passes) has two rounds of the optimiser, i.e. the OTHER ranks next enter a round
# hook (counts all-to-all), not the image exchange a cancelled rank used to run off to
who = int(sys.argv[4])
gpt.log = lambda rec: gpt.cancel() if (rec["iter"] == 2 and rank == who) else None
try:
    gpt.render()
    outcome = "finished"
except ppg_host.PPGError as ex:
    outcome = "ppg-error %d" % ex.code
except RenderAborted:
    outcome = "aborted"
done = [it["iter"] for it in gpt.iterations]
open(os.path.join(sys.argv[3], "cancel-rank%d.txt" % rank), "w").write("%s %s" % (outcome, done))
# a second render with the same reducer works: the status word does not stick
gpt.log = None
img = gpt.render()
open(os.path.join(sys.argv[3], "again-rank%d.txt" % rank), "w").write("%.6f" % float(img.mean()))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,who", [(2, 1), (3, 0)])
def test_cancelled_rank_takes_the_others_out_with_a_learned_fraction(oracle_lib, tmp_path, world, who):
    """with the round hook active (bsdfSamplingFractionLoss = kl) a cancelled rank used to leave renderPassesNoStat between two
    rounds and go to the image all-reduce while the others entered the next round's hook — different collectives on one communicator.  Now
    the library keeps it in step (empty rounds), every exchange begins with the sum of the ranks' status words, and all ranks leave the render
    at the same exchange — within the timeout, none hanging; the reducer is usable again afterwards."""
    script = tmp_path / "cancel_worker.py"
    script.write_text(CANCEL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(31500 + os.getpid() % 2000), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], str(script), os.path.join(ROOT, "practical-path-guiding_amd"),
           os.path.join(ROOT, "oracle", "libppg_oracle.so"), str(tmp_path), str(who)]
    subprocess.run(cmd, check=True, env=env, timeout=300, capture_output=True)
    outcomes = [(tmp_path / ("cancel-rank%d.txt" % r)).read_text() for r in range(world)]
    assert all(o.startswith(("ppg-error", "aborted")) and o.endswith("[0, 1, 2]") for o in outcomes), outcomes  # nobody finished, nobody got further
    again = {(tmp_path / ("again-rank%d.txt" % r)).read_text() for r in range(world)}
    assert len(again) == 1 and float(again.pop()) > 0.01


EARLY_CANCEL_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import ctypes, torch.distributed as dist
import ppg_host
from ppg_host.distributed import HostReducer, RenderAborted
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = ctypes.CDLL(sys.argv[2])
props = dict(budgetType=sys.argv[5], maxDepth=6, rrDepth=10, strictNormals=1, budget=(60 if sys.argv[5] == "spp" else 600.0), seed=17, sppPerPass=1,
             bsdfSamplingFractionLoss=sys.argv[6], sampleCombination="inversevar")
e = ppg_host.Engine(lib, "ppgo_", **props)
lib.ppgo_set_modes(e.ctx, 0, 0, 2)
e.set_scene(ppg_host.cbox_scene(48, 32)); e.set_shard(rank, world, 16)
gpt = ppg_host.GuidedPathTracer(engine=e, reducer=HostReducer(dist))
who = int(sys.argv[4])
if rank == who:
    gpt.cancel()  # BEFORE render(): the library's cancel is sticky and ppg_begin_render consumes it — this rank never reaches an exchange of the render itself
try:
    gpt.render()
    outcome = "finished"
except ppg_host.PPGError as ex:
    outcome = "ppg-error %d" % ex.code
except RenderAborted:
    outcome = "aborted"
open(os.path.join(sys.argv[3], "early-rank%d.txt" % rank), "w").write("%s %d" % (outcome, len(gpt.iterations)))
img = gpt.render() if sys.argv[5] == "spp" else None  # the reducer and the engine are usable again
open(os.path.join(sys.argv[3], "again-rank%d.txt" % rank), "w").write("%.6f" % (float(img.mean()) if img is not None else 1.0))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,who,budget,loss", [(2, 1, "spp", "kl"), (2, 0, "spp", "none"), (3, 2, "seconds", "none")])
def test_rank_cancelled_before_render_takes_the_others_out(oracle_lib, tmp_path, world, who, budget, loss):
    """ppg_cancel is sticky, so a cancel() that arrives before render() makes ppg_begin_render return PPG_ERR_CANCELLED — and that
    rank used to leave render() without having joined any exchange, its peers waiting for it in their first collective until the timeout.
    Every sharded render now begins with one status exchange (the all-reduce of stop_decision): the rank that cannot start says so there and
    all ranks leave together — before a single pass is rendered, whatever the budget type or the loss (the peers' first exchange would have
    been the image all-reduce, the stop hook or a round hook)."""
    script = tmp_path / "early_cancel_worker.py"
    script.write_text(EARLY_CANCEL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(33500 + os.getpid() % 2000), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], str(script), os.path.join(ROOT, "practical-path-guiding_amd"),
           os.path.join(ROOT, "oracle", "libppg_oracle.so"), str(tmp_path), str(who), budget, loss]
    subprocess.run(cmd, check=True, env=env, timeout=120, capture_output=True)
    outcomes = [(tmp_path / ("early-rank%d.txt" % r)).read_text() for r in range(world)]
    assert outcomes[who].startswith("ppg-error") and outcomes[who].endswith(" 0"), outcomes
    assert all(o == "aborted 0" for r, o in enumerate(outcomes) if r != who), outcomes
    again = {(tmp_path / ("again-rank%d.txt" % r)).read_text() for r in range(world)}
    assert len(again) == 1 and float(again.pop()) > 0.01


SECONDS_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import ctypes, numpy as np, torch.distributed as dist
import ppg_host
from ppg_host.distributed import HostReducer
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = ctypes.CDLL(sys.argv[2])
props = dict(budgetType="seconds", budget=1.2, maxDepth=6, rrDepth=10, strictNormals=1, seed=5, sampleCombination=sys.argv[4], sppPerPass=1,
             bsdfSamplingFractionLoss="kl" if sys.argv[4] == "inversevar" else "none")
e = ppg_host.Engine(lib, "ppgo_", **props)
lib.ppgo_set_modes(e.ctx, 0, 0, 1 + rank)  # ranks of different speed: their own clocks would disagree
e.set_scene(ppg_host.cbox_scene(32, 24)); e.set_shard(rank, world, 8)
gpt = ppg_host.GuidedPathTracer(engine=e, reducer=HostReducer(dist))
img = gpt.render()
rec = [[it["passes"], [s["passes_rendered_total"] for s in it["stats"]]] for it in gpt.iterations]
json.dump({"iterations": rec, "mean": float(np.nanmean(img)), "finite": bool(np.isfinite(img).all())}, open(os.path.join(sys.argv[3], "sec-rank%d.json" % rank), "w"))
np.save(os.path.join(sys.argv[3], "sec-rank%d.npy" % rank), img)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("combo", ["automatic", "inversevar"])
def test_time_budget_sharded_over_gloo(oracle_lib, tmp_path, combo):
    """budgetType = seconds — the reference's default — sharded over two ranks of different speed: every decision taken by a clock (the
    per-pass abort of performRenderPasses GP:1259-1262, the iteration loop and the switch to FINAL of renderTime GP:1434-1514) is rank 0's,
    broadcast (ppg_set_stop_hook + reducer.broadcast), so both ranks render the same iterations and the same number of passes in each,
    their collectives match, and both hold the same complete picture.  With a learned sampling fraction the round hooks stay in step too."""
    import json
    script = tmp_path / "seconds_worker.py"
    script.write_text(SECONDS_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(33500 + os.getpid() % 2000), OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], str(script), os.path.join(ROOT, "practical-path-guiding_amd"),
           os.path.join(ROOT, "oracle", "libppg_oracle.so"), str(tmp_path), combo]
    subprocess.run(cmd, check=True, env=env, timeout=300, capture_output=True)
    a, b = (json.load(open(tmp_path / ("sec-rank%d.json" % r))) for r in range(2))
    assert a["iterations"] == b["iterations"] and len(a["iterations"]) >= 3
    assert [it[0] for it in a["iterations"]] == [1 << k for k in range(len(a["iterations"]))]
    assert a["finite"] and b["finite"] and a["mean"] > 0.01
    assert np.array_equal(np.load(tmp_path / "sec-rank0.npy"), np.load(tmp_path / "sec-rank1.npy"))


FINAL8_WORKER = r'''
import os, sys, json
sys.path.insert(0, sys.argv[1])
import ctypes, numpy as np, torch.distributed as dist
import ppg_host
from ppg_host.distributed import HostReducer
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = ctypes.CDLL(sys.argv[2])
props = dict(budgetType="spp", maxDepth=6, rrDepth=10, strictNormals=1, budget=int(sys.argv[4]), seed=9, sppPerPass=1, sampleCombination="automatic",
             bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=400)
e = ppg_host.Engine(lib, "ppgo_", **props)
lib.ppgo_set_modes(e.ctx, 0, 0, 1)
e.set_scene(ppg_host.cbox_scene(32, 32)); e.set_shard(rank, world, 4)
gpt = ppg_host.GuidedPathTracer(engine=e, reducer=HostReducer(dist))
img = gpt.render()
last = gpt.iterations[-1]
json.dump({"passes": [it["passes"] for it in gpt.iterations], "final_samples": last["stats"][-1]["samples"]}, open(os.path.join(sys.argv[3], "f8-rank%d.json" % rank), "w"))
np.save(os.path.join(sys.argv[3], "f8-rank%d.npy" % rank), img)
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("budget,final_passes", [(20, 13), (127, 64)])
def test_final_iteration_of_few_groups_is_shared_by_eight_ranks(oracle_lib, tmp_path, budget, final_passes):
    """VERDICT r4 / ADVICE r4: a final iteration of 13 passes (the driver's `--steps 20`) is ONE group of passes, one of 64 passes four — dealt
    whole, one rank rendered most of the render alone while seven idled.  With fewer than two groups per rank every rank now renders every
    group on its own TILES (include/ppg.h "Final iteration: groups of passes"); the sums a pixel goes through are the same, so the picture
    still equals the single-rank one bit for bit, and every one of the eight ranks renders its share of the final samples."""
    import json
    import ppg_host
    world = 8
    script = tmp_path / "final8_worker.py"
    script.write_text(FINAL8_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(35500 + os.getpid() % 2000), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], str(script), os.path.join(ROOT, "practical-path-guiding_amd"),
           os.path.join(ROOT, "oracle", "libppg_oracle.so"), str(tmp_path), str(budget)]
    subprocess.run(cmd, check=True, env=env, timeout=900, capture_output=True)
    props = dict(budgetType="spp", maxDepth=6, rrDepth=10, strictNormals=1, budget=budget, seed=9, sppPerPass=1, sampleCombination="automatic",
                 bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box", sTreeThreshold=400)
    e = make_oracle(oracle_lib, threads=4, **props)
    e.set_scene(ppg_host.cbox_scene(32, 32)); e.render()
    ref = e.read_film()
    total = 32 * 32 * final_passes
    shares = []
    for r in range(world):
        got = json.load(open(tmp_path / ("f8-rank%d.json" % r)))
        assert got["passes"][-1] == final_passes
        shares.append(got["final_samples"])
        assert np.array_equal(np.load(tmp_path / ("f8-rank%d.npy" % r)), ref) and np.isfinite(ref).all() and ref.mean() > 0.01
    assert sum(shares) == total and min(shares) >= total // (2 * world), shares   # (64 tiles of 4 x 4 pixels over 8 ranks: an eighth each)


SECONDS_CANCEL_WORKER = r'''
import os, sys, threading
sys.path.insert(0, sys.argv[1])
import ctypes, torch.distributed as dist
import ppg_host
from ppg_host.distributed import HostReducer, RenderAborted
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lib = ctypes.CDLL(sys.argv[2])
props = dict(budgetType="seconds", budget=60.0, maxDepth=6, rrDepth=10, strictNormals=1, seed=5, sampleCombination=sys.argv[4], sppPerPass=1)  # loss = none: no round hook
e = ppg_host.Engine(lib, "ppgo_", **props)
lib.ppgo_set_modes(e.ctx, 0, 0, 1)
e.set_scene(ppg_host.cbox_scene(32, 24)); e.set_shard(rank, world, 8)
gpt = ppg_host.GuidedPathTracer(engine=e, reducer=HostReducer(dist))
if rank == 1:
    threading.Timer(1.5, gpt.cancel).start()   # in the middle of some iteration's passes: the others are in (or on their way to) a stop hook
try:
    gpt.render()
    outcome = "finished"
except ppg_host.PPGError as ex:
    outcome = "ppg-error %d" % ex.code
except RenderAborted:
    outcome = "aborted"
open(os.path.join(sys.argv[3], "sc-rank%d.txt" % rank), "w").write(outcome)
# the reducer and the engine are usable again: a short render finishes on every rank
e2 = ppg_host.Engine(lib, "ppgo_", **dict(props, budget=0.5))
lib.ppgo_set_modes(e2.ctx, 0, 0, 1)
e2.set_scene(ppg_host.cbox_scene(32, 24)); e2.set_shard(rank, world, 8)
gpt2 = ppg_host.GuidedPathTracer(engine=e2, reducer=gpt.reducer)
img = gpt2.render()
open(os.path.join(sys.argv[3], "sc-again-rank%d.txt" % rank), "w").write("%.6f" % float(img.mean()))
dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.parametrize("combo", ["automatic", "discard"])
def test_cancelled_rank_takes_the_others_out_of_a_time_budget(oracle_lib, tmp_path, combo):
    """budgetType = seconds without a learned fraction — the reference's defaults — has no round hook; a cancelled rank used to
    leave the batch loop for the image exchange while the others asked the stop hook (a broadcast without a status word): mismatched
    collectives.  Now the stop decision is ONE all-reduce of (rank 0's decision, status words), a cancelled rank asks it once more before it
    leaves, everybody stops there, and the image exchange that follows aborts the render on all ranks — long before the 60 s budget."""
    import time
    script = tmp_path / "seconds_cancel_worker.py"
    script.write_text(SECONDS_CANCEL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(37500 + os.getpid() % 2000), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], str(script), os.path.join(ROOT, "practical-path-guiding_amd"),
           os.path.join(ROOT, "oracle", "libppg_oracle.so"), str(tmp_path), combo]
    t0 = time.monotonic()
    subprocess.run(cmd, check=True, env=env, timeout=120, capture_output=True)
    assert time.monotonic() - t0 < 50   # (nobody rendered out the 60 s)
    outcomes = [(tmp_path / ("sc-rank%d.txt" % r)).read_text() for r in range(3)]
    assert all(o.startswith(("ppg-error", "aborted")) for o in outcomes), outcomes
    again = {(tmp_path / ("sc-again-rank%d.txt" % r)).read_text() for r in range(3)}
    assert len(again) == 1 and float(again.pop()) > 0.01


def test_final_partials_size_is_a_function_of_film_and_pass_count(oracle_lib):
    """What lets a rank that failed still join the final groups' collective (include/ppg.h): the float count of the exchange buffer follows from
    the film size and the pass count alone — 4 n + groups * 7 n, groups = ceil(passes / ppg_final_group_passes(passes)) — and is what the
    library hands out, whether the groups were dealt whole to the ranks or rendered by tiles."""
    import ppg_host
    for budget, world in ((20, 2), (45, 2), (300, 2)):
        props = dict(CBOX_PROPS, budget=budget, seed=1, sppPerPass=1, sampleCombination="discard")
        e = make_oracle(oracle_lib, threads=2, **props)
        e.set_scene(ppg_host.cbox_scene(16, 12)); e.set_shard(0, world, 4); e.begin_render()
        passes, it = 0, 0
        while passes < budget:  # renderSPP's schedule (GP:1367-1374)
            remaining = budget - passes
            n = min(remaining, 1 << it)
            if remaining - n < 2 * n:
                n = remaining
            final = n >= remaining
            e.begin_iteration(final)
            e.render_passes_nostat(n)
            if final:
                ptr, count = e.final_partials()
                g = oracle_lib.ppgo_final_group_passes(n)
                assert count == e.final_partials_expected(n) == 4 * 192 + (-(-n // g)) * 7 * 192 and ptr
                e.final_partials_commit()
            e.finish_passes(); e.build_sdtree(); e.end_iteration()
            passes += n; it += 1


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
