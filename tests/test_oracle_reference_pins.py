"""Pin (2) of the oracle: the reference's own shipped outputs (tests/golden/ref_logs.json and
ref_cbox_images.npz, mined from scenes/*/*.exr by tools/make_ref_fixtures.py).

The reference's sampler streams are not reproducible (per-thread SFMT, OS-scheduled blocks), so these are
statistical pins with stated tolerances; the schedule pins are exact."""
import json
import math
import os

import numpy as np
import pytest

from conftest import CBOX_PROPS, GOLDEN, make_oracle


@pytest.fixture(scope="module")
def ref_logs():
    return json.load(open(os.path.join(GOLDEN, "ref_logs.json")))["scenes"]


def _schedule(budget, spp):
    """renderSPP's pass schedule (GP:1352-1374) as the oracle's Python mirror computes it."""
    n_passes = int(math.ceil(budget / float(spp)))
    out, done, it = [], 0, 0
    while done < n_passes:
        rem = n_passes - done
        p = min(rem, 1 << it)
        if rem - p < 2 * p:
            p = rem
        out.append(p); done += p; it += 1
    return out


@pytest.mark.parametrize("scene,budget,spp", [("cbox", 127, 4), ("cbox-improved", 127, 1), ("kitchen", 1020, 4),
                                              ("kitchen-improved", 2400, 1), ("spaceship", 1024, 4), ("spaceship-improved", 1023, 1)])
def test_iteration_schedule_matches_reference_logs(ref_logs, scene, budget, spp):
    logged = [it["passes"] for it in ref_logs[scene]["iterations"]]
    if scene == "kitchen":  # that log was cut by its 'automatic' FINAL branch; compare the common prefix
        assert _schedule(2400, 4)[:len(logged) - 1] == logged[:-1]
    else:
        assert _schedule(budget, spp) == logged


@pytest.fixture(scope="module")
def cbox_run(oracle_lib):
    import ppg_host
    e = make_oracle(oracle_lib, threads=os.cpu_count() or 8, budget=127, seed=20240926, **CBOX_PROPS)
    gpt = ppg_host.GuidedPathTracer(engine=e)
    img = gpt.render(ppg_host.cbox_scene(512, 512))
    return gpt.iterations, img


def test_cbox_iteration_statistics_match_reference_log(cbox_run, ref_logs):
    its, _ = cbox_run
    ref = ref_logs["cbox"]["iterations"]
    assert [i["passes"] for i in its] == [r["passes"] for r in ref] == [1, 2, 4, 8, 17]
    # iteration 0: one 85-node depth-4 D-tree, 4.15 recorded vertices per path, mean radiance 0.1357
    t0 = its[0]["tree"]
    assert (t0["min_nodes"], t0["max_nodes"], t0["min_depth"], t0["max_depth"], t0["n_leaves"]) == (85, 85, 4, 4, 1)
    assert abs(t0["avg_stat_weight"] / ref[0]["stat_weight"][1] - 1) < 0.005       # 4 349 763 in the log
    assert abs(t0["avg_mean_radiance"] / ref[0]["mean_radiance"][1] - 1) < 0.05   # 0.135707 (noisy estimator)
    # iteration 1: refine(thr = 16970) -> 512 leaves sharing one topology
    t1 = its[1]["tree"]
    assert t1["n_leaves"] == 512 and t1["min_nodes"] == t1["max_nodes"] and abs(t1["min_nodes"] - 68) <= 5
    for k in (1, 2, 3):
        t, r = its[k]["tree"], ref[k]
        assert abs(t["avg_stat_weight"] / r["stat_weight"][1] - 1) < 0.03
        assert abs(t["max_stat_weight"] / r["stat_weight"][2] - 1) < 0.03
        assert abs(t["avg_mean_radiance"] / r["mean_radiance"][1] - 1) < 0.06
        assert abs(t["avg_depth"] - r["depth"][1]) < 0.1
    for k in (2, 3, 4):
        assert abs(its[k]["tree"]["avg_nodes"] - ref[k]["node_count"][1]) < 1.5
    # per-iteration variance estimate (GP:1300-1313)
    for k in range(5):
        assert abs(its[k]["stats"][0]["variance"] / ref[k]["var"][0] - 1) < 0.15
    # "Average path length : 5.20 (174.48 M rays / 33.55 M samples)"
    rays = sum(s["rays"] for i in its for s in i["stats"]); samples = sum(s["samples"] for i in its for s in i["stats"])
    plen = sum(s["path_length_sum"] for i in its for s in i["stats"])
    assert samples == 512 * 512 * 128 and abs(samples / ref_logs["cbox"]["samples"] - 1) < 1e-3
    assert abs(rays / samples - ref_logs["cbox"]["rays"] / ref_logs["cbox"]["samples"]) < 0.03
    assert abs(plen / samples - ref_logs["cbox"]["avg_path_length"]) < 0.03


def test_cbox_image_matches_reference_render(cbox_run):
    _, img = cbox_run
    ref = np.load(os.path.join(GOLDEN, "ref_cbox_images.npz"))
    assert np.allclose(img.mean((0, 1)), ref["cbox_mean_rgb"], rtol=0.01)  # [0.4451, 0.1581, 0.0333]
    blocks = img.reshape(64, 8, 64, 8, 3).mean((1, 3))
    rmse = np.sqrt(((blocks - ref["cbox_block8"]) ** 2).mean())
    noise_floor = np.sqrt(((ref["cbox_improved_block8"] - ref["cbox_block8"]) ** 2).mean())  # two reference renders
    assert rmse < 1.6 * noise_floor, (rmse, noise_floor)
    coarse = blocks.reshape(8, 8, 8, 8, 3).mean((1, 3)) / ref["cbox_block8"].reshape(8, 8, 8, 8, 3).mean((1, 3))
    assert np.abs(coarse - 1).max() < 0.08  # no spatial or per-channel bias (geometry, BSDFs, emitter, camera)


def test_cbox_improved_preset_matches_reference_log(oracle_lib, ref_logs):
    """The "improved" configuration (scenes/cbox/cbox-improved.xml: inverse-variance combination, KL-learned BSDF sampling fraction,
    stochastic spatial + box directional filter, sTreeThreshold 4000, 1 spp per pass) against the log embedded in the reference's
    cbox-improved.exr.  Iteration 1 — the first guided pass — is excluded from the tight bounds: the reference steps Adam after every
    ~2 records in arrival order, the oracle's default rule steps 64 mini-batches per pass from exact sums, so the
    learned fraction lags there (variance 6.2 vs 4.7, 15 % fewer records); from iteration 2 on the statistics agree."""
    import ppg_host
    from conftest import IMPROVED
    ref = ref_logs["cbox-improved"]["iterations"]
    e = make_oracle(oracle_lib, threads=os.cpu_count() or 8, budget=127, seed=20240926, **dict(CBOX_PROPS, **IMPROVED))
    gpt = ppg_host.GuidedPathTracer(engine=e)
    img = gpt.render(ppg_host.cbox_scene(512, 512))
    its = gpt.iterations
    assert [i["passes"] for i in its] == [r["passes"] for r in ref] == [1, 2, 4, 8, 16, 32, 64]
    t0 = its[0]["tree"]
    assert (t0["min_nodes"], t0["max_nodes"], t0["min_depth"], t0["max_depth"], t0["n_leaves"]) == (85, 85, 4, 4, 1)
    assert abs(t0["avg_stat_weight"] / ref[0]["stat_weight"][1] - 1) < 0.005   # 1 088 232 records of the unguided first pass
    assert its[1]["tree"]["n_leaves"] == 512 and its[1]["tree"]["min_nodes"] == its[1]["tree"]["max_nodes"] == 68  # refine(1.09 M, thr 2000); reset topology
    for k in (2, 3, 4, 5):
        t, r = its[k]["tree"], ref[k]
        assert abs(t["avg_stat_weight"] / r["stat_weight"][1] - 1) < 0.06, k
        assert abs(t["avg_depth"] - r["depth"][1]) < 0.1 and abs(t["avg_nodes"] - r["node_count"][1]) < 1.5, k
        assert abs(t["avg_mean_radiance"] / r["mean_radiance"][1] - 1) < 0.08, k
    for k in (2, 3, 4, 5, 6):
        assert abs(its[k]["stats"][0]["variance"] / ref[k]["var"][0] - 1) < 0.15, k
    assert abs(its[6]["tree"]["avg_depth"] - ref[6]["depth"][1]) < 0.1 and abs(its[6]["tree"]["avg_nodes"] - ref[6]["node_count"][1]) < 1.5
    samples = sum(s["samples"] for i in its for s in i["stats"]); plen = sum(s["path_length_sum"] for i in its for s in i["stats"])
    assert samples == 512 * 512 * 127 and abs(plen / samples - ref_logs["cbox-improved"]["avg_path_length"]) < 0.15   # 6.49 in the log
    gold = np.load(os.path.join(GOLDEN, "ref_cbox_images.npz"))
    assert np.allclose(img.mean((0, 1)), gold["cbox_improved_mean_rgb"], rtol=0.012)
    blocks = img.reshape(64, 8, 64, 8, 3).mean((1, 3))
    rmse = np.sqrt(((blocks - gold["cbox_improved_block8"]) ** 2).mean())
    noise_floor = np.sqrt(((gold["cbox_improved_block8"] - gold["cbox_block8"]) ** 2).mean())
    assert rmse < 1.6 * noise_floor, (rmse, noise_floor)


def test_sampling_fraction_rules_against_the_reference_log_seed_averaged(oracle_lib, ref_logs):
    """The learned BSDF sampling fraction shows in the variance estimates of cbox-improved.exr's log.  The reference's numbers are one
    draw each; the oracle's need not be — averaged over seeds their spread (2-3 % per run) drops below the effect measured here.

    (1) PPGO_ADAM_SEQUENTIAL, the reference's literal rule (a step after every ~2 records, in arrival order — single-threaded here so the
        order is defined): iteration 1, the first guided passes and the ones most sensitive to the optimiser, reproduces the log's 4.713.
    (2) The product's rule (PPGO_ADAM_ROUND: the same append()/step() arithmetic applied at the end of each round of passes in key order,
        fractions frozen within a round — include/ppg.h) starts every pass of iteration 1 from the untrained fraction 0.5: +19 % there,
        a lag that is gone two iterations later.  That deviation is what this test measures; DESIGN.md §4.4 quotes it."""
    import ppg_host
    from conftest import IMPROVED
    ref = [r["var"][0] for r in ref_logs["cbox-improved"]["iterations"]]
    assert abs(ref[1] - 4.713) < 1e-3 and abs(ref[3] - 0.297) < 1e-3
    scene = ppg_host.cbox_scene(512, 512)

    def variances(adam, budget, seeds):
        rows = []
        for seed in seeds:
            e = make_oracle(oracle_lib, threads=os.cpu_count() or 8, adam=adam, budget=budget, seed=seed, **dict(CBOX_PROPS, **IMPROVED))
            gpt = ppg_host.GuidedPathTracer(engine=e)
            gpt.render(scene)
            rows.append([i["stats"][0]["variance"] for i in gpt.iterations])
        return np.mean(np.array(rows)[:, 1:], axis=0)

    seq = variances(1, 7, (1000, 1001, 1002, 1003))              # iterations 0-2 (the last one does not train: 7, not 3, passes)
    assert abs(seq[0] / ref[1] - 1) < 0.05, seq                   # measured 4.757 +- 0.04 (six seeds) vs 4.713
    rnd = variances(0, 31, (1000, 1001, 1002))                    # iterations 0-4
    assert 1.08 < rnd[0] / ref[1] < 1.32, rnd                     # measured 5.63: the documented lag of the round rule
    assert abs(rnd[1] / ref[2] - 1) < 0.20, rnd                   # 1.071 vs 0.922 (the literal rule: 0.976)
    assert abs(rnd[2] / ref[3] - 1) < 0.07 and abs(rnd[3] / ref[4] - 1) < 0.09, rnd   # 0.290 vs 0.297, 0.0881 vs 0.0834: converged


SPACESHIP = "/root/reference/scenes/spaceship/spaceship.xml"
IMPROVED_PRESET = dict(sampleCombination="inversevar", bsdfSamplingFractionLoss="kl", spatialFilter="stochastic", directionalFilter="box",
                       sTreeThreshold=4000, sppPerPass=1)


@pytest.mark.skipif(not os.path.exists(SPACESHIP), reason="reference scenes not mounted (development container only)")
@pytest.mark.parametrize("variant", ["spaceship", "spaceship-improved"])
def test_spaceship_scene_matches_the_reference_render(oracle_lib, variant):
    """The reference's bundled SPACESHIP scene end to end: its XML through ppg_host.load_scene (86 OBJ meshes — two are missing from the
    checkout and skipped —, twosided rough conductors, rough plastic cut from Mitsuba's data/microfacet tables, a rough dielectric canopy,
    four area lights and the emitting sky sphere with flipped normals), rendered by the oracle at the scene's 640 x 360 with 63 spp,
    against the pixels of the reference's own 1023-spp render (scenes/spaceship/spaceship.exr; likewise spaceship-improved.xml / .exr, the README's improved preset).  Region means agree to Monte-Carlo
    noise: 0.0188 on the backdrop is (sky 0.3 x albedo 0.1) x visibility — it was 0.033 while the loader still gave the emitting dome
    Mitsuba's 0.5 "convenience" BSDF instead of the all-absorbing one of Shape::configure (shape.cpp:48-72)."""
    import sys
    import ppg_host
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import exr_min
    _, ch = exr_min.read_exr(os.path.join(os.path.dirname(SPACESHIP), variant + ".exr"))
    ref = np.stack([ch[k] for k in ("R", "G", "B")], -1)
    desc, props, info = ppg_host.load_scene(os.path.join(os.path.dirname(SPACESHIP), variant + ".xml"), strict=False, data_dir="/root/reference/mitsuba/data")
    assert len(info["warnings"]) == 2 and all("not found" in w for w in info["warnings"])
    assert desc.n_triangles == 257486 and len(desc.spheres) == 1 and len(desc.emitters) == 5 and desc.rtrans.shape == (3, 101)
    base = dict(strictNormals=1, maxDepth=10, rrDepth=10, budgetType="spp", budget=1023.0)
    assert props == (base if variant == "spaceship" else dict(base, **IMPROVED_PRESET))  # spaceship-improved.xml = README.md:30-37
    e = make_oracle(oracle_lib, threads=os.cpu_count() or 8, **dict(props, budget=63.0))
    img = ppg_host.GuidedPathTracer(engine=e).render(desc)
    assert img.shape == ref.shape == (360, 640, 3)
    regions = dict(backdrop=(0, 80, 0, 200, 0.01), floor=(300, 360, 0, 640, 0.01), right=(100, 200, 560, 640, 0.02), ship=(100, 260, 200, 500, 0.04),
                   whole=(0, 360, 0, 640, 0.02))
    for name, (y0, y1, x0, x1, tol) in regions.items():
        a, b = np.nanmean(img[y0:y1, x0:x1].reshape(-1, 3), 0), ref[y0:y1, x0:x1].reshape(-1, 3).mean(0)
        assert np.allclose(a, b, rtol=tol), (name, a, b)


def _logged(variant):
    import json
    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_logs.json")))["scenes"][variant]["iterations"]


@pytest.mark.skipif(not os.path.exists(SPACESHIP), reason="reference scenes not mounted (development container only)")
def test_spaceship_tree_statistics_follow_the_reference_log(oracle_lib):
    """the reference's render log of scenes/spaceship/spaceship.exr prints, per iteration, the SD-tree statistics of
    GP:1176-1186 — depth, mean radiance, node count, statistical weight, each as [min, avg, max] — and the variance estimate.  The oracle on
    the same XML at the same 640 x 360 reproduces them iteration by iteration (two of the 86 meshes are missing from the checkout: the
    recorded vertices of the first pass come out 0.2 % low).  Tolerances = three times the spread over seeds 0 / 3 / 5 or the stated
    systematic offset, whichever is larger; a wrong BSDF lobe, MIS weight or refinement rule moves these numbers by far more."""
    import ppg_host
    desc, props, _ = ppg_host.load_scene(SPACESHIP, strict=False, data_dir="/root/reference/mitsuba/data")
    log = _logged("spaceship")
    e = make_oracle(oracle_lib, threads=os.cpu_count() or 8, **dict(props, budget=60.0, seed=3))  # 15 passes of 4 spp: iterations 0, 1, 2 + a final one
    g = ppg_host.GuidedPathTracer(engine=e)
    g.render(desc)
    t = [it["tree"] for it in g.iterations]
    var = [it["stats"][-1]["variance"] for it in g.iterations]
    # seed sweep (0 / 3 / 5 / 7 / 11, round 5): statistical weights within 0.25 % of the log in every iteration (first pass: -0.2 % systematic,
    # the missing meshes), their maxima within 0.6 %, leaves exact, average depth within 0.02, nodes within 2; the average of the leaves' mean
    # radiance within 1.5 % except for one leaf with a firefly (seed 3, iteration 1: +10 %); the variance estimate within 8 % in iterations 0 - 1
    # (one outlier of +35 %) — and in iteration 2 consistently HALF the log's 0.040083, which itself breaks the log's own 1 / N sequence
    # (0.0364, 0.0398, 0.0401, 0.0082): one heavy-tailed draw of the reference, not compared.
    assert abs(t[0]["avg_stat_weight"] / log[0]["stat_weight"][1] - 1) < 0.005                       # 1 843 434 vs 1 847 293 recorded vertices
    assert abs(t[0]["avg_mean_radiance"] / log[0]["mean_radiance"][1] - 1) < 0.02                    # 0.1259 vs 0.126198
    assert abs(var[0] / log[0]["var"][0] - 1) < 0.08                                                  # 0.0359 - 0.0376 vs 0.036389
    assert t[1]["n_leaves"] == 128 and t[2]["n_leaves"] == 253 and t[1]["max_depth"] == 5 == int(log[1]["depth"][2])
    for k in (1, 2):
        assert abs(t[k]["avg_stat_weight"] / log[k]["stat_weight"][1] - 1) < 0.008, (k, t[k])         # 18 449 vs 18 415; 19 058 vs 19 100
        assert abs(t[k]["max_stat_weight"] / log[k]["stat_weight"][2] - 1) < 0.02, (k, t[k])          # 632 303 vs 630 512
        assert abs(t[k]["avg_mean_radiance"] / log[k]["mean_radiance"][1] - 1) < 0.15, (k, t[k])      # 0.01369 vs 0.013705 (a firefly leaf: +10 %)
        assert abs(t[k]["avg_nodes"] - log[k]["node_count"][1]) < 3 and abs(t[k]["avg_depth"] - log[k]["depth"][1]) < 0.05, (k, t[k])
    assert abs(var[1] / log[1]["var"][0] - 1) < 0.45, var                                             # 0.0396 - 0.0429 (0.0537 once) vs 0.039828


@pytest.mark.skipif(not os.path.exists(SPACESHIP), reason="reference scenes not mounted (development container only)")
def test_spaceship_improved_literal_adam_rule_follows_the_reference_log_and_the_round_rule_lags(oracle_lib):
    """The improved preset on a real scene (spaceship-improved.xml: KL-learned sampling fraction, stochastic + box filters).  With the
    LITERAL rule — every record applied to the optimiser when its path ends, gradient at the variable's value at that moment
    (PPGO_ADAM_SEQUENTIAL, single-threaded so the order is defined) — the oracle reproduces the reference's log iteration by iteration:
    average / maximum statistical weight within 2 %, variance estimate within 4 % (measured in round 5, seed 3, six iterations: variance
    0.1002 / 0.0420 / 0.0186 / 0.00803 / 0.00364 vs the log's 0.0976 / 0.0399 / 0.0180 / 0.00805 / 0.00362).  The arithmetic is the reference's.

    The ROUND rule of product and oracle (include/ppg.h: fractions frozen during a round) does not: the literal rule's variable follows the
    records of the image region being rendered — it rises to 1.1 - 1.8 while a leaf's directly visible surfaces are rendered and falls back
    below 0.6 by the end of the pass — which a per-leaf value frozen for a pass cannot.  Measured (seeds 3 / 4 / 5): the variance estimate is
    1.8 - 2.2 x the log's in iteration 1, 1.6 - 2.5 x in iterations 2 - 4, 1.1 - 1.4 x in iteration 5 and equal from iteration 6 (64
    passes) on; 17 % fewer recorded vertices in iteration 1 (paths sampled with a lower BSDF fraction end sooner).  This is the price of
    rendering a pass as ONE wavefront of a million paths instead of 16 threads' worth; it is stated, not hidden."""
    import ppg_host
    desc, props, _ = ppg_host.load_scene(os.path.join(os.path.dirname(SPACESHIP), "spaceship-improved.xml"), strict=False, data_dir="/root/reference/mitsuba/data")
    log = _logged("spaceship-improved")

    def run(adam, threads, budget):
        e = make_oracle(oracle_lib, threads=threads, adam=adam, **dict(props, budget=budget, seed=3))
        g = ppg_host.GuidedPathTracer(engine=e)
        g.render(desc)
        return [it["tree"] for it in g.iterations], [it["stats"][-1]["variance"] for it in g.iterations]

    t, var = run(1, 1, 15.0)      # literal rule: iterations 0, 1, 2 + a final one
    assert abs(t[0]["avg_stat_weight"] / log[0]["stat_weight"][1] - 1) < 0.005                       # 460 396 vs 462 239
    for k in (1, 2):
        assert abs(t[k]["avg_stat_weight"] / log[k]["stat_weight"][1] - 1) < 0.03, (k, t[k])          # 2911.6 vs 2867.0; 3065.6 vs 3042.8
        assert abs(t[k]["max_stat_weight"] / log[k]["stat_weight"][2] - 1) < 0.03, (k, t[k])          # 130 589 vs 129 329; 128 065 vs 126 901
        assert abs(var[k] / log[k]["var"][0] - 1) < 0.1, (k, var[k])                                  # 0.1002 vs 0.0976; 0.0420 vs 0.0399
    t, var = run(0, os.cpu_count() or 8, 15.0)      # round rule
    assert 1.5 < var[1] / log[1]["var"][0] < 2.6 and 0.78 < t[1]["avg_stat_weight"] / log[1]["stat_weight"][1] < 0.88, (var, t[1])   # 0.2003 vs 0.0976; 2368 vs 2867


@pytest.mark.skipif(not os.path.exists(SPACESHIP), reason="reference scenes not mounted (development container only)")
def test_spaceship_improved_region_rounds_reproduce_the_reference_log(oracle_lib):
    """Round 6, the experiment the mechanism of the test above suggests: if the reference's variable tracks the image REGION
    being rendered, rounds by region should reproduce it.  PPGO_ADAM_REGIONS + R (oracle only): every pass of the early iterations is rendered
    in R groups of 32 x 32 blocks, consecutive in the spiral order of the reference's block scheduler (imageproc.cpp:29-80), the optimiser
    applied after every group.  Measured on spaceship-improved, variance estimate of iterations 1 - 4, seeds 3 / 4 (profiles/r06_experiments.json):
        reference log            0.0976          0.0399          0.0180          0.00805
        round rule (shipped)     0.200  0.211    0.063  0.098    0.087  0.058    0.0180 0.0214
        R = 4                    0.172  0.183    0.067  0.091    0.049  0.031    0.0094 0.0089
        R = 8                    0.110  0.112    0.046  0.068    0.024  0.020    0.0084 0.0082
        R = 16                   0.105  0.104    0.045  0.055    0.0194 0.0193   0.0082 0.0081
        R = 32                   0.102  0.100    0.043  0.045    0.0191 0.0189   0.0082 0.0081
    — the more regions, the closer to the log; with 16 per pass within 4 - 13 % (one noisy draw of iteration 2 apart), average statistical weight
    per leaf 2754 / 3121 / 3678 / 4994 against the log's 2867 / 3043 / 3767 / 4951.  The lag of the round rule IS the missing regional feedback
    and nothing else.  The product does not render this way: 16 rounds per pass in iterations 1 - 4 are 480 rounds of 60 - 900 k paths instead
    of 8, each with its own tail, sorts and host round trips, in the regime where one MI355X runs at a third of its throughput (DESIGN.md
    section 7: batches below ~2 M paths) — estimated at +150 ms or more on every KITCHEN render at 1280 x 720, more than the driver's whole
    20-pass command, for the benefit of the first 30 passes.  The test pins the ORACLE's region mode against the
    reference: one more independent check of its Adam arithmetic, filters and statistics."""
    import ppg_host
    desc, props, _ = ppg_host.load_scene(os.path.join(os.path.dirname(SPACESHIP), "spaceship-improved.xml"), strict=False, data_dir="/root/reference/mitsuba/data")
    log = _logged("spaceship-improved")
    e = make_oracle(oracle_lib, threads=os.cpu_count() or 8, adam=16 + 16, **dict(props, budget=15.0, seed=3))  # PPGO_ADAM_REGIONS + 16
    g = ppg_host.GuidedPathTracer(engine=e)
    g.render(desc)
    t, var = [it["tree"] for it in g.iterations], [it["stats"][-1]["variance"] for it in g.iterations]
    for k in (1, 2):
        assert abs(var[k] / log[k]["var"][0] - 1) < 0.2, (k, var)                                        # 0.1048 vs 0.0976; 0.0450 vs 0.0399
        assert abs(t[k]["avg_stat_weight"] / log[k]["stat_weight"][1] - 1) < 0.06, (k, t[k])              # 2754 vs 2867; 3121 vs 3043


KITCHEN = "/root/reference/scenes/kitchen/kitchen-improved.xml"


@pytest.mark.skipif(not os.path.exists(KITCHEN), reason="reference scenes not mounted (development container only)")
def test_kitchen_scene_matches_the_reference_render(oracle_lib):
    """The reference's bundled KITCHEN scene (BASELINE.json's headline configuration) end to end: kitchen-improved.xml through
    ppg_host.load_scene — 283 of its 289 OBJ meshes (six are missing from the checkout), 63 BSDFs incl. rough plastic with bitmap textures
    on the diffuse reflectance (eleven JPEG / PNG files, texture coordinates, UV tangents), and its only light, the `sunsky` emitter, baked
    into a 512 x 256 radiance map from the Hosek-Wilkie / Preetham tables of the Mitsuba tree (ppg_host/sunsky.py) — rendered by the oracle
    at a quarter of the scene's 700 x 400 with 63 spp, against the pixels of the reference's own 2400-spp render
    (scenes/kitchen/kitchen-improved.exr).  The whole-image mean agrees to 1-2 % per channel ([0.634, 0.677, 0.728] in the reference), means over a 4 x 4 grid
    of regions to Monte-Carlo noise + the missing meshes.  A wrong sun position, sky scale, sRGB decoding, v-flip or UV interpolation
    each moves these numbers by far more."""
    import ppg_host
    from ppg_host import imageio
    ref = imageio.read_image(os.path.join(os.path.dirname(KITCHEN), "kitchen-improved.exr"))
    desc, props, info = ppg_host.load_scene(KITCHEN, strict=False, width=175, height=100, data_dir="/root/reference/mitsuba/data")
    assert len(info["warnings"]) == 6 and all("not found" in w for w in info["warnings"])  # nothing else is skipped or substituted
    assert desc.n_triangles == 1021815 and len(desc.textures) == 11 and desc.envmap is not None and desc.envmap["rgb"].shape == (256, 512, 3)
    assert props == dict(strictNormals=1, budgetType="spp", budget=2400.0, **IMPROVED_PRESET)
    e = make_oracle(oracle_lib, threads=os.cpu_count() or 8, **dict(props, budget=63.0))
    gpt = ppg_host.GuidedPathTracer(engine=e)
    img = gpt.render(desc)
    assert np.isfinite(img).all()
    a, b = img.reshape(-1, 3).mean(0), ref.reshape(-1, 3).mean(0)
    assert np.allclose(a, b, rtol=0.03), (a, b)
    lum = np.array([0.212671, 0.715160, 0.072169])
    R = ref[:400, :700].reshape(4, 100, 4, 175, 3).mean((1, 3)) @ lum
    O = img.reshape(4, 25, 4, 43, 3).mean((1, 3)) @ lum if False else img[:100, :172].reshape(4, 25, 4, 43, 3).mean((1, 3)) @ lum
    assert np.abs(O / R - 1).max() < 0.3 and np.abs(O / R - 1).mean() < 0.12, O / R
    samples = sum(s["samples"] for it in gpt.iterations for s in it["stats"]); plen = sum(s["path_length_sum"] for it in gpt.iterations for s in it["stats"])
    assert 5.5 < plen / samples < 8.5  # 6.44 in the reference's log (2400 spp; the early, unguided iterations weigh more at 63 spp)
    # the reference's log of this render: vertices recorded by the first pass (statistical weight of the one D-tree of iteration 0), per pixel
    log = json.load(open(os.path.join(GOLDEN, "ref_logs.json")))["scenes"]["kitchen-improved"]
    per_pixel_ref = log["iterations"][0]["stat_weight"][1] / (log["width"] * log["height"])
    per_pixel = gpt.iterations[0]["tree"]["max_stat_weight"] / (175 * 100)
    print("recorded vertices per pixel, first pass:", per_pixel, "reference log:", per_pixel_ref)
    assert abs(per_pixel / per_pixel_ref - 1) < 0.015, (per_pixel, per_pixel_ref)  # measured 4.5621 vs 4.5596 (+0.05 %); 17 500 paths: noise ~0.5 %
