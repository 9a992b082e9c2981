"""The BENCHMARKED scenes in the driver's GPU test run: the reference's own KITCHEN (scenes/kitchen/kitchen-improved.xml — the workload of
bench.py's line, BASELINE.json configs[2]) and SPACESHIP (configs[3]) as converted in the development container into scratch/*.ppgs
(`python -m ppg_host <scene>.xml --lenient --data-dir <mitsuba>/data --ppgs ...`; the reference tree cannot travel, its scene DATA does —
the files ship to the GPU box with the repository snapshot but are too large for the history, so these tests skip where they are absent).

  * GPU = oracle, bit for bit (film, SD-tree topology and sums, ray / vertex counters), on the real geometry, BSDF mix, textures and
    baked sunsky at a film size the CPU restatement covers in seconds;
  * at the film sizes of the configurations (1280x720, 1920x1080): the properties that need no oracle — run-to-run determinism of film
    and SD-tree, sample / ray bookkeeping, a finite film — on the exact file bench.py times.
"""
import copy
import os

import numpy as np
import pytest

from conftest import IMPROVED, ROOT, make_oracle

pytestmark = pytest.mark.gpu

KITCHEN = os.path.join(ROOT, "scratch", "kitchen-improved.ppgs")
SPACESHIP = os.path.join(ROOT, "scratch", "spaceship.ppgs")


def _props(path, **over):
    from bench import scene_props
    p = scene_props(path, dict(budgetType="spp", seed=1234))
    p.update(over)
    return p


def _load(path, w, h):
    import ppg_host
    scene = ppg_host.load_scene_file(path)
    scene = copy.copy(scene)
    scene.camera = ppg_host.resize_camera(scene.camera, w, h)  # keeps the horizontal field of view
    return scene


def _stats(gpt):
    return [[s["samples"], s["rays"], s["path_length_sum"], s["vertices_committed"]] for it in gpt.iterations for s in it["stats"]]


def _assert_tree_equal(a, b):
    assert np.array_equal(a["children"], b["children"]) and np.array_equal(a["axis"], b["axis"])
    for k in ("sampling", "building"):
        for f in a[k]:
            assert np.array_equal(np.asarray(a[k][f]), np.asarray(b[k][f])), (k, f)


@pytest.mark.skipif(not os.path.exists(KITCHEN), reason="scratch/kitchen-improved.ppgs (conversion of the reference's KITCHEN scene) not present")
def test_kitchen_improved_against_oracle(oracle_lib):
    """configs[2] on its real inputs: 1 021 815 triangles, 63 BSDFs (rough plastic / conductor, glass, thin glass, mirror, two-sided
    diffuse), eleven bitmap textures, the baked sunsky environment map; improved preset, unbounded depth; 160x90, 31 spp (5 iterations,
    four rounds of the optimiser)."""
    import ppg_host
    from test_gpu_parity import assert_tree_equal, hip
    scene = _load(KITCHEN, 160, 90)
    assert scene.n_triangles > 1000000 and len(scene.textures) >= 10 and scene.envmap is not None
    props = _props(KITCHEN, budget=31.0)
    assert props["bsdfSamplingFractionLoss"] == "kl" and props["sampleCombination"] == "inversevar" and int(props.get("maxDepth", -1)) == -1
    g, o = hip(**props), make_oracle(oracle_lib, threads=os.cpu_count() or 8, **props)
    gg, go = ppg_host.GuidedPathTracer(engine=g), ppg_host.GuidedPathTracer(engine=o)
    ig, io = gg.render(scene), go.render(scene)
    assert [i["passes"] for i in gg.iterations] == [1, 2, 4, 8, 16]
    assert _stats(gg) == _stats(go)
    assert np.array_equal(ig, io, equal_nan=True) and np.isfinite(ig).all() and ig.mean() > 1e-3
    assert_tree_equal(g.read_sdtree(), o.read_sdtree())


@pytest.mark.skipif(not os.path.exists(SPACESHIP), reason="scratch/spaceship.ppgs (conversion of the reference's SPACESHIP scene) not present")
def test_spaceship_improved_against_oracle(oracle_lib):
    """configs[3] on its real inputs (257 486 triangles, the emitting sky sphere, rough conductors / rough plastic / rough dielectric
    canopy) with spaceship-improved.xml's settings (the improved preset, maxDepth 10, rrDepth 10); 320x180, 31 spp."""
    import ppg_host
    from test_gpu_parity import assert_tree_equal, hip
    scene = _load(SPACESHIP, 320, 180)
    assert scene.n_triangles > 250000 and len(scene.spheres) == 1
    props = _props(SPACESHIP, budget=31.0, **IMPROVED)
    assert int(props["maxDepth"]) == 10 and int(props["rrDepth"]) == 10
    g, o = hip(**props), make_oracle(oracle_lib, threads=os.cpu_count() or 8, **props)
    gg, go = ppg_host.GuidedPathTracer(engine=g), ppg_host.GuidedPathTracer(engine=o)
    ig, io = gg.render(scene), go.render(scene)
    assert _stats(gg) == _stats(go)
    assert np.array_equal(ig, io, equal_nan=True) and np.isfinite(ig).all() and ig.mean() > 1e-3
    assert_tree_equal(g.read_sdtree(), o.read_sdtree())


@pytest.mark.parametrize("which", ["kitchen-720p", "spaceship-1080p"])
def test_real_scenes_full_size_invariants(which):
    """The files bench.py times, at the film sizes of BASELINE.json configs[2] / configs[3]: two renders give the same film and the same
    SD-tree bit for bit (the sub-batches, side streams and generations of the persistent-thread tail must not show); samples, rays and
    path lengths add up; the film is finite; nothing is recorded in the final iteration."""
    import ppg_host
    from test_gpu_parity import assert_tree_equal, hip
    path, (w, h), budget, extra = {"kitchen-720p": (KITCHEN, (1280, 720), 31.0, {}),
                                   "spaceship-1080p": (SPACESHIP, (1920, 1080), 31.0, IMPROVED)}[which]
    if not os.path.exists(path):
        pytest.skip("%s not present" % os.path.relpath(path, ROOT))
    scene = _load(path, w, h)
    props = _props(path, budget=budget, **extra)
    runs = []
    for _ in range(2):
        e = hip(**props)
        gpt = ppg_host.GuidedPathTracer(engine=e)
        img = gpt.render(scene)
        assert sum(i["passes"] for i in gpt.iterations) == int(budget)
        for i in gpt.iterations:
            st = i["stats"][0]
            assert st["samples"] == w * h * i["passes"] and st["rays"] >= st["path_length_sum"]  # (+ the look-through rays behind null surfaces)
            assert (st["vertices_committed"] > 0) == (i is not gpt.iterations[-1])
        assert np.isfinite(img).all() and img.mean() > 1e-3
        runs.append((img, e.read_sdtree(), _stats(gpt)))
    assert runs[0][2] == runs[1][2]
    assert np.array_equal(runs[0][0], runs[1][0])
    assert_tree_equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("which", ["spaceship", "kitchen-improved", "kitchen"])
def test_tree_statistics_follow_the_reference_logs(which):
    """The reference's render logs (embedded in scenes/*/*.exr; tests/golden/ref_logs.json) print the SD-tree statistics of GP:1176-1186 after
    every iteration.  This build, on the converted scene files at the reference's own film sizes, reproduces them iteration by iteration
    (tolerances: the spread of four seeds, tools/ref_log_probe.py, round 5):

    SPACESHIP (default settings, 640 x 360): recorded vertices of the first pass 1 842 413 vs 1 847 293 (two meshes are missing from the
    checkout), then average statistical weight per leaf within 0.4 %, its maximum within 1.2 %, 128 / 253 / 447 / 690 leaves exactly, average
    depth within 0.03, node count within 1 (from iteration 2), mean radiance within 2 % (from iteration 2), variance estimate within 30 %.
    KITCHEN (improved preset, 700 x 400; six meshes missing): recorded vertices of the first pass within 0.4 %; from iteration 2 on average
    statistical weight within 12 %, depth within 0.12, node count within 1.5, variance estimate within 25 % — and iteration 1's statistical weight
    14 - 21 % LOW: the round rule of the sampling-fraction optimiser, the one stated deviation, visible where it lags.
    KITCHEN with the DEFAULT settings (kitchen.xml = the same scene without the improved preset: 4 spp per pass, nearest filters, no learned
    fraction — so no round rule either): recorded vertices of the first pass 5 120 637 vs 5 106 572 (+0.3 %), average statistical weight per
    leaf within 5 % in iterations 1 - 4 (10 410 - 10 860 vs 10 880; 44 955 - 45 323 vs 45 565), depth within 0.04, node count within 2 from
    iteration 2 on.  That log's VARIANCE estimates are not compared: 696 / 297 / 8.4 in iterations 0 - 2 are incompatible with the improved
    log of the same scene under the formula of GP:1300-1313 (a directly visible sky of radiance ~7 contributes ~2.5 to iteration 0's
    estimate at 4 spp, which is what this build measures: 1.96 - 2.01) — the shipped kitchen.exr predates something."""
    import json
    import ppg_host
    from test_gpu_parity import hip
    path = SPACESHIP if which == "spaceship" else KITCHEN
    if not os.path.exists(path):
        pytest.skip("scene file not present")
    log = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_logs.json")))["scenes"][which]
    scene = _load(path, log["width"], log["height"])
    props = _props(path, seed=1234) if which != "kitchen" else dict(budgetType="spp", strictNormals=1, seed=1234)  # kitchen.xml:4-18
    spp = int(props.get("sppPerPass", 4))
    e = hip(**dict(props, budget=float(63 * spp)))
    gpt = ppg_host.GuidedPathTracer(engine=e)
    gpt.render(scene)
    assert [it["passes"] for it in gpt.iterations] == [1, 2, 4, 8, 16, 32]
    t = [it["tree"] for it in gpt.iterations]
    var = [it["stats"][-1]["variance"] for it in gpt.iterations]
    ref = log["iterations"]
    rel = lambda a, b: abs(a / b - 1)
    assert rel(t[0]["avg_stat_weight"], ref[0]["stat_weight"][1]) < 0.006 and t[0]["n_leaves"] == 1 and t[0]["avg_nodes"] == 85.0
    if which == "spaceship":
        assert [x["n_leaves"] for x in t[1:5]] == [128, 253, 447, 690]
        assert rel(var[0], ref[0]["var"][0]) < 0.08
        for k in (1, 2, 3, 4):
            assert rel(t[k]["avg_stat_weight"], ref[k]["stat_weight"][1]) < 0.006, (k, t[k])
            assert rel(t[k]["max_stat_weight"], ref[k]["stat_weight"][2]) < 0.02, (k, t[k])
            assert abs(t[k]["avg_depth"] - ref[k]["depth"][1]) < 0.04 and abs(t[k]["avg_nodes"] - ref[k]["node_count"][1]) < (5 if k == 1 else 1.5), (k, t[k])
            if k >= 2:
                assert rel(t[k]["avg_mean_radiance"], ref[k]["mean_radiance"][1]) < 0.03, (k, t[k])
            if k >= 3:
                assert rel(var[k], ref[k]["var"][0]) < 0.4, (k, var)      # (iteration 2 of the log breaks its own 1 / N sequence: one heavy-tailed draw)
    elif which == "kitchen":
        assert t[1]["n_leaves"] == 512
        for k in (1, 2, 3, 4):
            assert rel(t[k]["avg_stat_weight"], ref[k]["stat_weight"][1]) < 0.08, (k, t[k])
            assert abs(t[k]["avg_depth"] - ref[k]["depth"][1]) < (1.01 if k == 1 else 0.08), (k, t[k])   # (iteration 1: one integer depth for all leaves, 6 or 7)
            if k >= 2:
                assert abs(t[k]["avg_nodes"] - ref[k]["node_count"][1]) < 2.5 and rel(t[k]["avg_mean_radiance"], ref[k]["mean_radiance"][1]) < 0.25, (k, t[k])
    else:
        assert 0.75 < t[1]["avg_stat_weight"] / ref[1]["stat_weight"][1] < 0.9, t[1]      # the round rule's lag (2728 - 2983 vs 3466)
        for k in (2, 3, 4):
            assert rel(t[k]["avg_stat_weight"], ref[k]["stat_weight"][1]) < 0.15, (k, t[k])
            assert abs(t[k]["avg_depth"] - ref[k]["depth"][1]) < 0.2 and abs(t[k]["avg_nodes"] - ref[k]["node_count"][1]) < 2, (k, t[k])
        for k in (3, 4, 5):
            assert rel(var[k], ref[k]["var"][0]) < 0.3, (k, var)


def test_kitchen_default_configuration_picture_matches_the_reference():
    """KITCHEN with the DEFAULT settings of scenes/kitchen/kitchen.xml (strictNormals, spp budget 2400; everything else the plug-in's
    defaults: 4 spp per pass, nearest filters, no learned fraction, sampleCombination = automatic, maxDepth = -1), pinned by its PICTURE — until
    round 6 this configuration on the real scene was pinned by tree statistics only (test_tree_statistics_follow_the_reference_logs; its log's
    variance column is not comparable).  At the reference's 700 x 400, per seed (tools/kitchen_default_probe.py, three seeds
    on the MI355X: profiles/r06_kitchen_default_picture.json):
      * the error against the reference's converged kitchen-reference.exr over the pixels outside the six missing meshes' footprint is the
        error of the reference's OWN render of this configuration, kitchen.exr: MAPE 0.0934, RMSE 0.115 (tests/golden/ref_kitchen_reference.npz)
        — within 8 %;
      * the 50 x 50-pixel block means agree with kitchen.exr's, block by block, wherever no missing mesh shows."""
    import ppg_host
    from test_gpu_parity import hip
    if not os.path.exists(KITCHEN):
        pytest.skip("scene file not present")
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ref_kitchen_reference.npz"))
    ref = fx["rgb"].astype(np.float64)
    blk = int(fx["mask_block"])
    keep = ~np.kron(fx["mask_blocks"], np.ones((blk, blk), np.uint8)).astype(bool)
    keep50 = keep.reshape(8, 50, 14, 50).all((1, 3))
    assert keep50.sum() >= 60
    scene = _load(KITCHEN, 700, 400)
    for seed in (1234, 98765):
        e = hip(budgetType="spp", strictNormals=1, budget=2400.0, seed=seed)  # kitchen.xml:4-18
        gpt = ppg_host.GuidedPathTracer(engine=e)
        img = gpt.render(scene).astype(np.float64)
        assert sum(it["passes"] + it.get("final_passes", 0) for it in gpt.iterations) == 600
        d = (img - ref)[keep]
        mape, rmse = float((np.abs(d) / (ref[keep] + 0.01)).mean()), float(np.sqrt((d * d).mean()))
        # measured, three seeds: MAPE 0.0940 / 0.0992 / 0.0972 against the reference's own 0.0934; RMSE (a few fireflies' worth either way) 0.120 /
        # 0.152 / 0.146 against 0.115; block means of the 71 blocks clear of the missing meshes within 0.9 - 1.3 % of kitchen.exr's on average,
        # 7.8 - 12.8 % in the worst block (two renders of 2400 spp each, the default configuration's fireflies in either)
        assert abs(mape / float(fx["kitchen_mape_unmasked"]) - 1) < 0.08, (seed, mape, float(fx["kitchen_mape_unmasked"]))
        assert rmse < 1.5 * float(fx["kitchen_rmse_unmasked"]), (seed, rmse)
        b50 = img.reshape(8, 50, 14, 50, 3).mean((1, 3)).mean(-1)
        rel = np.abs(b50 / fx["kitchen_blocks50"].astype(np.float64).mean(-1) - 1)
        assert rel[keep50].mean() < 0.02 and rel[keep50].max() < 0.2, (seed, float(rel[keep50].mean()), float(rel[keep50].max()))


def test_region_rounds_follow_the_reference_log_of_spaceship_improved():
    """The PRODUCT against the reference's own log where the shipped round rule is known to lag (a14): spaceship-improved
    (the bundled scene with the improved preset, 640 x 360) with `ppg_set_adam_regions(16)` — the variance estimate of iterations 1 - 3 follows
    the log (0.0976 / 0.0399 / 0.0180) within the spread the oracle showed for this rule (+7 / +13 / +8 %, seeds 3 and 4), where the default
    rounds give 2.0 / 1.6 - 2.5 / 3 - 5 x; average statistical weight per leaf within 6 %."""
    import json
    import ppg_host
    from test_gpu_parity import hip
    if not os.path.exists(SPACESHIP):
        pytest.skip("scene file not present")
    log = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_logs.json")))["scenes"]["spaceship-improved"]
    scene = _load(SPACESHIP, log["width"], log["height"])
    props = dict(_props(SPACESHIP, seed=3), budget=31.0, sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic",
                 directionalFilter="box", sTreeThreshold=4000, sppPerPass=1)   # spaceship-improved.xml = spaceship.xml + the improved preset
    ref = log["iterations"]

    def run(regions):
        e = hip(**props)
        e.set_adam_regions(regions)
        g = ppg_host.GuidedPathTracer(engine=e)
        g.render(scene)
        assert [it["passes"] for it in g.iterations] == [1, 2, 4, 8, 16]
        return [it["stats"][-1]["variance"] for it in g.iterations], [it["tree"]["avg_stat_weight"] for it in g.iterations]

    var, sw = run(16)
    for k in (1, 2, 3):
        assert abs(var[k] / ref[k]["var"][0] - 1) < 0.25, (k, var)
        assert abs(sw[k] / ref[k]["stat_weight"][1] - 1) < 0.06, (k, sw)
    var0, _ = run(0)
    assert var0[1] / ref[1]["var"][0] > 1.5   # the default rounds: the lag this option removes


def test_stragglers_invariants_at_the_benchmark_size(monkeypatch):
    """The stragglers' machinery at the bench's full size (KITCHEN scene-improved, 1280 x 720, 20 passes), where the oracle cannot follow:
    scheduling-only splits do not show in any result (the final iteration in two launches with stragglers handed over at depth 8; one stream and no
    split at all: picture, SD-tree, fractions and counters of the default schedule bit for bit), a render with a THIRD of all paths as stragglers
    (the tests' switch: depth 8) is deterministic from run to run, and it differs from the default one from the first round on only."""
    import ctypes as C
    import ppg_host
    from test_gpu_parity import hip
    if not os.path.exists(KITCHEN):
        pytest.skip("scene file not present")
    scene = ppg_host.load_scene_file(KITCHEN)
    props = _props(KITCHEN, budget=20.0, seed=1234)

    def run(env, depth=0):
        with monkeypatch.context() as m:
            for k, v in env.items():
                m.setenv(k, v)
            e = hip(**props)
        if depth:
            e._call("debug_set_defer_depth", C.c_int32(depth))
        g = ppg_host.GuidedPathTracer(engine=e)
        img = g.render(scene)
        t = e.read_sdtree()
        stats = np.array([[s["rays"], s["path_length_sum"], s["vertices_committed"], s["samples"]] for it in g.iterations for s in it["stats"]], np.uint64)
        return img, t["theta"], t["children"], t["sampling"]["node_sums"], stats

    def same(a, b):
        return all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))

    base = run({})
    assert [int(v) for v in base[4][:, 3]] == [1280 * 720 * n for n in (1, 2, 4, 13)]
    assert same(base, run(dict(PPG_FINAL_HALVES="1", PPG_SPLIT_DEPTH="8")))
    assert same(base, run(dict(PPG_SPLIT_DEPTH="0", PPG_NO_OVERLAP="1")))
    d8 = run({}, 8)
    assert same(d8, run({}, 8))
    assert not np.array_equal(d8[1], base[1]) and np.array_equal(d8[4][:1], base[4][:1]) and np.array_equal(d8[4][:, 3], base[4][:, 3])
