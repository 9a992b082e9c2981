"""The BSDF plug-ins of SURVEY.md §8(f1) as restated in the oracle, checked the way the reference checks its own BSDFs
(M/src/tests/test_chisquare.cpp: the sampling routine must be distributed like pdf(), sample() must return eval / pdf):
  * pdf integrates to (at most) one over the sphere, equal to the non-delta sampling probability,
  * the weight returned by sample() equals eval() / pdf() at the sampled direction,
  * a histogram of sampled directions matches the integral of pdf() per bin,
  * reciprocity / energy conservation where the model has them, Fresnel limits, Snell's law.
"""
import ctypes as C

import numpy as np
import pytest

from ppg_host.bindings import Material

FP = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(FP)


def _slice(lib, mat):
    """roughplastic: hand the oracle's element-wise BSDF interface the material's rough-transmittance slice (tests/golden/rtrans_slices.npz)."""
    if "slice" in mat:
        sl = np.ascontiguousarray(mat["slice"], np.float32)
        assert lib.ppgo_bsdf_set_rtrans(_fp(sl), len(sl) - 1) == 0


def rtrans_slice(distribution, alpha, eta):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rtrans_slices.npz"))
    for i, (d, a, e) in enumerate(g["cases"]):
        if (str(d), float(a), float(e)) == (distribution, alpha, eta):
            return g["slice%d" % i]
    raise KeyError((distribution, alpha, eta))


def roughplastic(distribution, alpha, eta, **kw):
    return dict(type="roughplastic", alpha=alpha, eta=eta, distribution=distribution, slice=rtrans_slice(distribution, alpha, eta), **kw)


def bsdf_eval(lib, mat, wi, wo):
    _slice(lib, mat)
    m = Material.from_dict(mat)
    wi = np.ascontiguousarray(np.broadcast_to(wi, wo.shape), np.float32); wo = np.ascontiguousarray(wo, np.float32)
    f = np.zeros_like(wo); pdf = np.zeros(len(wo), np.float32)
    assert lib.ppgo_bsdf_eval(C.byref(m), len(wo), _fp(wi), _fp(wo), _fp(f), _fp(pdf)) == 0
    return f, pdf


def bsdf_sample(lib, mat, wi, xy):
    _slice(lib, mat)
    m = Material.from_dict(mat)
    n = len(xy)
    wi = np.ascontiguousarray(np.broadcast_to(wi, (n, 3)), np.float32); xy = np.ascontiguousarray(xy, np.float32)
    wo = np.zeros((n, 3), np.float32); w = np.zeros((n, 3), np.float32)
    pdf = np.zeros(n, np.float32); eta = np.zeros(n, np.float32); delta = np.zeros(n, np.int32)
    assert lib.ppgo_bsdf_sample(C.byref(m), n, _fp(wi), _fp(xy), _fp(wo), _fp(w), _fp(pdf), _fp(eta), delta.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    return wo, w, pdf, eta, delta


def sphere_grid(n_theta=1800, n_phi=720):
    """Midpoint grid in (theta, phi) — fine enough in theta for a GGX lobe of alpha = 0.05: directions + solid-angle weights."""
    th = (np.arange(n_theta) + 0.5) / n_theta * np.pi
    ph = (np.arange(n_phi) + 0.5) / n_phi * 2 * np.pi
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(TH) * np.cos(PH), np.sin(TH) * np.sin(PH), np.cos(TH)], -1).reshape(-1, 3)
    dw = (np.sin(TH) * (np.pi / n_theta) * (2 * np.pi / n_phi)).reshape(-1)
    return d.astype(np.float32), dw, (n_theta, n_phi)


def unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v)).astype(np.float32)


GOLD = dict(type="roughconductor", alpha=0.25, eta=(0.143, 0.375, 1.442), k=(3.983, 2.386, 1.603), reflectance=(1, 1, 1))
SMOOTH = [
    ("diffuse", dict(type="diffuse", reflectance=(0.2, 0.5, 0.8))),
    ("twosided-diffuse", dict(type="diffuse", reflectance=(0.2, 0.5, 0.8), twosided=True)),
    ("ggx-0.25", GOLD),
    ("ggx-0.05", dict(GOLD, alpha=0.05)),
    ("ggx-0.6-twosided", dict(GOLD, alpha=0.6, twosided=True)),
    ("ggx-beckmann-0.25", dict(GOLD, distribution="beckmann")),
    ("ggx-beckmann-0.08", dict(GOLD, alpha=0.08, distribution="beckmann")),
    ("ggx-beckmann-0.5", dict(GOLD, alpha=0.5, distribution="beckmann")),
    ("ggx-roughdielectric-0.3", dict(type="roughdielectric", alpha=0.3, eta=1.5, reflectance=(1, 1, 1), specular=(0.9, 0.95, 1.0))),
    ("ggx-roughdielectric-beckmann-0.15", dict(type="roughdielectric", alpha=0.15, eta=1.33, reflectance=(1, 1, 1), specular=(1, 1, 1), distribution="beckmann")),
    ("ggx-roughplastic-0.1", roughplastic("ggx", 0.1, 1.5, reflectance=(0.6, 0.3, 0.1), specular=(1, 1, 1))),
    ("ggx-roughplastic-beckmann-0.2-nonlinear-twosided", roughplastic("beckmann", 0.2, 1.49, reflectance=(0.2, 0.5, 0.7), specular=(0.8, 0.9, 1.0),
                                                                    nonlinear=True, twosided=True)),
    ("plastic", dict(type="plastic", reflectance=(0.6, 0.3, 0.1), specular=(1, 1, 1), eta=1.49)),
    ("plastic-nonlinear", dict(type="plastic", reflectance=(0.6, 0.3, 0.1), specular=(0.8, 0.8, 0.8), eta=1.9, nonlinear=True)),
]


@pytest.mark.parametrize("name,mat", SMOOTH, ids=[n for n, _ in SMOOTH])
@pytest.mark.parametrize("wi", [(0, 0, 1), (0.6, 0.2, 0.5), (-0.3, 0.9, 0.08), (0.5, -0.3, -0.6)], ids=["normal", "oblique", "grazing", "from-below"])
def test_sampling_matches_pdf_and_weight_matches_eval(oracle_lib, name, mat, wi):
    wi = unit(wi)
    rng = np.random.RandomState(7)
    xy = rng.rand(400000, 2).astype(np.float32)
    wo, w, pdf, eta, delta = bsdf_sample(oracle_lib, mat, wi, xy)
    ok = (pdf > 0) & (w.sum(1) > 0)
    if wi[2] < 0 and not ("twosided" in name or "roughdielectric" in name):
        assert not ok.any()                      # one-sided BRDFs are black from behind
        return
    smooth = ok & (delta == 0)
    if "roughdielectric" in name:                # transmitted samples carry the relative IOR for Russian roulette
        trans = ok & (wo[:, 2] * wi[2] < 0)
        e_mat = float(np.float32(mat["eta"]))
        assert trans.any() and np.allclose(eta[trans], e_mat if wi[2] > 0 else 1 / e_mat, rtol=1e-6) and np.all(eta[ok & ~trans] == 1)
    else:
        assert np.all(eta[ok] == 1)
    # (1) weight == eval / pdf for the smooth component.  sample() returns the weight of the sampled component over the
    # probability of choosing it and the direction; eval()/pdf() of the solid-angle measure describe the same component.
    f, p = bsdf_eval(oracle_lib, mat, wi, wo[smooth])
    assert np.all(p > 0)
    assert np.allclose(w[smooth], f / p[:, None], rtol=2e-3, atol=1e-6)
    assert np.allclose(pdf[smooth], p, rtol=2e-3, atol=1e-7)
    # (2) pdf integrates to the probability of producing a smooth sample (1 for diffuse / GGX up to masked normals)
    d, dw, shape = sphere_grid()
    fg, pg = bsdf_eval(oracle_lib, mat, wi, d)
    if "roughdielectric" in name:
        # roughdielectric.cpp:352-417: pdf() does not test the refraction geometry (eval() does, through Smith's G), so it is positive on
        # a few percent of directions that sample() never produces; compare on the directions the model can actually scatter into
        pg = np.where(fg.sum(1) > 0, pg, 0).astype(np.float32)
    total = float((pg.astype(np.float64) * dw).sum())
    frac_smooth = float((delta[pdf > 0] == 0).mean()) if name.startswith("plastic") else 1.0
    if name.startswith("ggx"):
        assert 0.6 < total <= 1.005  # visible-normal sampling: reflections below the horizon are discarded, never > 1
        if "roughdielectric" in name:
            assert abs(total - ok.mean()) < 0.01
    else:
        assert abs(total - frac_smooth) < 0.01
    # (3) histogram of sampled directions vs. integral of the pdf per bin (coarse 10 x 20 bins)
    nb_t, nb_p = 18, 12
    H = (pg.astype(np.float64) * dw).reshape(shape).reshape(nb_t, shape[0] // nb_t, nb_p, shape[1] // nb_p).sum((1, 3))
    ws = wo[smooth].astype(np.float64)
    it = np.clip((np.arccos(np.clip(ws[:, 2], -1, 1)) / np.pi * nb_t).astype(int), 0, nb_t - 1)
    ip = np.clip(((np.arctan2(ws[:, 1], ws[:, 0]) % (2 * np.pi)) / (2 * np.pi) * nb_p).astype(int), 0, nb_p - 1)
    cnt = np.zeros((nb_t, nb_p)); np.add.at(cnt, (it, ip), 1)
    emp = cnt / len(xy)  # fraction of ALL draws landing in the bin as a smooth sample
    expect = H * (1.0 if not name.startswith("ggx") else 1.0)
    big = expect > 2e-3
    assert big.sum() >= 3
    assert np.allclose(emp[big], expect[big], rtol=0.08, atol=2e-4), (np.abs(emp[big] / expect[big] - 1).max())


def test_ggx_reciprocity_and_white_furnace(oracle_lib):
    mat = dict(GOLD, eta=(0, 0, 0), k=(1, 1, 1))  # Fresnel = 1: a perfectly reflecting rough mirror
    rng = np.random.RandomState(3)
    a = rng.randn(2000, 3); a[:, 2] = np.abs(a[:, 2]) + 0.05; a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.randn(2000, 3); b[:, 2] = np.abs(b[:, 2]) + 0.05; b /= np.linalg.norm(b, axis=1, keepdims=True)
    fab, _ = bsdf_eval(oracle_lib, mat, a.astype(np.float32), b.astype(np.float32))
    fba, _ = bsdf_eval(oracle_lib, mat, b.astype(np.float32), a.astype(np.float32))
    # eval returns f * cos(theta_o): f(a, b) cos_b / cos_b == f(b, a) cos_a / cos_a
    assert np.allclose(fab[:, 0] / b[:, 2], fba[:, 0] / a[:, 2], rtol=2e-3, atol=1e-6)
    # albedo <= 1 (single scattering loses energy for rough surfaces) and equal to the textbook closed forms
    # D = a^2 / (pi (cos^2 (a^2 - 1) + 1)^2), G1 = 2 / (1 + sqrt(1 + a^2 tan^2)) evaluated independently in double
    d, dw, _ = sphere_grid()
    wi = unit((0.2, 0.1, 0.97))
    for alpha in (0.05, 0.25, 0.6, 1.0):
        f, _ = bsdf_eval(oracle_lib, dict(mat, alpha=alpha), wi, d)
        albedo = float((f[:, 0].astype(np.float64) * dw).sum())
        up = d[:, 2] > 0
        wo = d[up].astype(np.float64)
        h = wo + wi.astype(np.float64); h /= np.linalg.norm(h, axis=1, keepdims=True)
        D = alpha ** 2 / (np.pi * (h[:, 2] ** 2 * (alpha ** 2 - 1) + 1) ** 2)
        g1 = lambda v: 2 / (1 + np.sqrt(1 + alpha ** 2 * (1 - v[..., 2] ** 2) / v[..., 2] ** 2))  # noqa: E731
        want = float((D * g1(wo) * g1(wi.astype(np.float64)) / (4 * wi[2]) * dw[up]).sum())
        assert abs(albedo - want) < 2e-4 * want and albedo <= 1.0, (alpha, albedo, want)


def test_conductor_fresnel_limits(oracle_lib):
    xy = np.array([[0.3, 0.3]], np.float32)
    # material "none" (eta = 0, k = 1): exactly 1 at every angle (conductor.cpp:171-173)
    for wi in ((0, 0, 1), (0.7, 0, 0.2), (0.3, -0.9, 0.01)):
        wo, w, pdf, eta, delta = bsdf_sample(oracle_lib, dict(type="mirror", reflectance=(0.9, 0.8, 0.7)), unit(wi), xy)
        assert np.array_equal(w[0], np.float32([0.9, 0.8, 0.7])) and delta[0] == 1 and pdf[0] == 1
        assert np.allclose(wo[0], unit(wi) * [-1, -1, 1])
    # gold at normal incidence: R = ((n - 1)^2 + k^2) / ((n + 1)^2 + k^2)
    n, k = np.array(GOLD["eta"]), np.array(GOLD["k"])
    wo, w, *_ = bsdf_sample(oracle_lib, dict(type="conductor", eta=GOLD["eta"], k=GOLD["k"], reflectance=(1, 1, 1)), unit((0, 0, 1)), xy)
    assert np.allclose(w[0], ((n - 1) ** 2 + k ** 2) / ((n + 1) ** 2 + k ** 2), rtol=1e-5)
    # grazing: reflectance → 1; from below: nothing
    _, w, *_ = bsdf_sample(oracle_lib, dict(type="conductor", eta=GOLD["eta"], k=GOLD["k"]), unit((1, 0, 1e-4)), xy)
    assert np.all(w[0] > 0.995)
    _, w, pdf, *_ = bsdf_sample(oracle_lib, dict(type="conductor", eta=GOLD["eta"], k=GOLD["k"]), unit((0.3, 0, -0.5)), xy)
    assert np.all(w[0] == 0) and pdf[0] == 0


def test_dielectric_snell_fresnel_and_radiance_scaling(oracle_lib):
    eta = 1.5
    mat = dict(type="dielectric", eta=eta, reflectance=(1, 1, 1), specular=(1, 1, 1))
    rng = np.random.RandomState(5)
    xy = rng.rand(200000, 2).astype(np.float32)
    for wi in (unit((0.5, 0.1, 0.8)), unit((0.5, 0.1, -0.8)), unit((0.9, 0, -0.3))):
        wo, w, pdf, e, delta = bsdf_sample(oracle_lib, mat, wi, xy)
        assert np.all(delta == 1)
        refl = wo[:, 2] * wi[2] > 0
        ci = abs(float(wi[2]))
        rel = eta if wi[2] > 0 else 1 / eta           # n_t / n_i
        s2 = (1 - ci * ci) / rel ** 2
        if s2 >= 1:  # total internal reflection
            assert refl.all() and np.allclose(pdf, 1)
            continue
        ct = np.sqrt(1 - s2)
        Rs = (ci - rel * ct) / (ci + rel * ct); Rp = (rel * ci - ct) / (rel * ci + ct)
        F = 0.5 * (Rs ** 2 + Rp ** 2)
        assert abs(refl.mean() - F) < 0.004 and np.allclose(pdf[refl], F, atol=1e-6) and np.allclose(pdf[~refl], 1 - F, atol=1e-6)
        assert np.allclose(wo[refl], wi * [-1, -1, 1], atol=1e-6) and np.all(e[refl] == 1) and np.allclose(w[refl], 1)
        t = wo[~refl][0]
        assert abs(np.linalg.norm(t) - 1) < 1e-5
        # Snell: sin_t = sin_i / rel, opposite side, same azimuth reversed
        assert abs(np.sqrt(t[0] ** 2 + t[1] ** 2) - np.sqrt(1 - ci * ci) / rel) < 1e-5 and t[2] * wi[2] < 0
        assert np.allclose(t[:2] / np.linalg.norm(t[:2]), -wi[:2] / np.linalg.norm(wi[:2]), atol=1e-5)
        # radiance scaling (n_i / n_t)^2 and the relative-IOR bookkeeping used by Russian roulette (GP:2040, 2130)
        assert np.allclose(w[~refl], (1 / rel) ** 2, rtol=1e-5) and np.allclose(e[~refl], rel, rtol=1e-6)
    f, p = bsdf_eval(oracle_lib, mat, unit((0.5, 0.1, 0.8)), sphere_grid(60, 120)[0])
    assert not f.any() and not p.any()  # delta BSDF: nothing in the solid-angle measure


def test_plastic_energy_and_delta_split(oracle_lib):
    mat = dict(type="plastic", reflectance=(0.9, 0.9, 0.9), specular=(1, 1, 1), eta=1.5)
    rng = np.random.RandomState(9)
    xy = rng.rand(300000, 2).astype(np.float32)
    for wi in (unit((0, 0, 1)), unit((0.8, 0, 0.25))):
        wo, w, pdf, e, delta = bsdf_sample(oracle_lib, mat, wi, xy)
        spec = delta == 1
        assert np.allclose(wo[spec], wi * [-1, -1, 1], atol=1e-6)
        albedo = w[:, 0].astype(np.float64).mean()  # E[weight] = hemispherical reflectance
        assert 0.5 < albedo < 1.0
        # the delta lobe carries exactly the Fresnel reflectance: E[weight; delta] = F
        ci = float(wi[2]); ct = np.sqrt(1 - (1 - ci * ci) / 1.5 ** 2)
        F = 0.5 * (((ci - 1.5 * ct) / (ci + 1.5 * ct)) ** 2 + ((1.5 * ci - ct) / (1.5 * ci + ct)) ** 2)
        assert abs((w[:, 0] * spec).astype(np.float64).mean() - F) < 0.004


def test_flags(oracle_lib):
    def flags(mat):
        m = Material.from_dict(mat)
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        oracle_lib.ppgo_bsdf_flags(C.byref(m), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value
    assert flags(dict(type="diffuse")) == (1, 0, 0) and flags(dict(type=1)) == (1, 0, 1)
    assert flags(dict(type="mirror")) == (0, 1, 0) and flags(dict(type="conductor", twosided=True)) == (0, 1, 1)
    assert flags(GOLD) == (1, 0, 0) and flags(dict(type="plastic")) == (1, 0, 0) and flags(dict(type="dielectric")) == (0, 1, 1)


def test_roughplastic_tends_to_smooth_plastic(oracle_lib):
    """roughplastic.cpp:330-501 with a small alpha is plastic.cpp with its delta lobe widened a little: the two albedos agree (the
    diffuse base sees the rough-transmittance table where the smooth plug-in sees 1 - Fresnel), and no more energy leaves than arrives."""
    rng = np.random.RandomState(11)
    xy = rng.rand(400000, 2).astype(np.float32)
    rough = roughplastic("ggx", 0.05, 1.9, reflectance=(0.6, 0.3, 0.1), specular=(1, 1, 1))
    smooth = dict(type="plastic", reflectance=(0.6, 0.3, 0.1), specular=(1, 1, 1), eta=1.9)
    for wi in (unit((0, 0, 1)), unit((0.6, 0.2, 0.5)), unit((0.9, 0, 0.3))):
        _, wr, pr, _, dr = bsdf_sample(oracle_lib, rough, wi, xy)
        _, ws, ps, _, _ = bsdf_sample(oracle_lib, smooth, wi, xy)
        ar, asm = wr.astype(np.float64).mean(0), ws.astype(np.float64).mean(0)
        assert np.all(dr == 0) and np.all(ar < 1.0) and np.allclose(ar, asm, rtol=0.03, atol=0.012), (ar, asm)
    m = Material.from_dict(rough)
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    _slice(oracle_lib, rough)
    oracle_lib.ppgo_bsdf_flags(C.byref(m), C.byref(a), C.byref(b), C.byref(c))
    assert (a.value, b.value, c.value) == (1, 0, 0)  # glossy + diffuse, one-sided


def test_thindielectric_null_component(oracle_lib):
    eta = 1.5
    mat = dict(type="thindielectric", eta=eta, reflectance=(1, 1, 1), specular=(0.9, 0.95, 1.0))
    rng = np.random.RandomState(6)
    xy = rng.rand(200000, 2).astype(np.float32)
    for wi in (unit((0, 0, 1)), unit((0.6, 0.1, -0.5)), unit((0.95, 0, 0.1))):
        wo, w, pdf, e, delta = bsdf_sample(oracle_lib, mat, wi, xy)
        ci = abs(float(wi[2])); ct = np.sqrt(1 - (1 - ci * ci) / eta ** 2)
        R = 0.5 * (((ci - eta * ct) / (ci + eta * ct)) ** 2 + ((eta * ci - ct) / (eta * ci + ct)) ** 2)
        Rp = R + (1 - R) ** 2 * R / (1 - R * R)  # = 2R / (1 + R): all internal bounces of the slab
        assert abs(Rp - 2 * R / (1 + R)) < 1e-12
        refl = wo[:, 2] * wi[2] > 0
        assert np.all(delta == 1) and np.all(e == 1)
        assert abs(refl.mean() - Rp) < 0.004 and np.allclose(pdf[refl], Rp, atol=1e-6) and np.allclose(pdf[~refl], 1 - Rp, atol=1e-6)
        assert np.allclose(wo[refl], wi * [-1, -1, 1], atol=1e-7) and np.allclose(wo[~refl], -wi, atol=1e-7)  # straight through
        assert np.allclose(w[refl], 1) and np.allclose(w[~refl], np.float32([0.9, 0.95, 1.0]))
    m = Material.from_dict(mat)
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    oracle_lib.ppgo_bsdf_flags(C.byref(m), C.byref(a), C.byref(b), C.byref(c))
    assert (a.value, b.value, c.value) == (0, 1, 1)


def test_mask_adapter(oracle_lib):
    """mask.cpp:108-214 around a diffuse BRDF: pass-through with probability 1 - luminance(opacity) and weight (1 - opacity) / (1 - prob);
    otherwise the nested sample scaled by opacity / prob; eval / pdf of the solid-angle measure scaled by opacity / prob."""
    op = np.float32([0.9, 0.6, 0.3])
    base = dict(type="diffuse", reflectance=(0.5, 0.6, 0.7))
    mat = dict(base, opacity=tuple(op))
    prob = float(np.float32(op[0] * np.float32(0.212671) + op[1] * np.float32(0.715160) + op[2] * np.float32(0.072169)))
    rng = np.random.RandomState(8)
    xy = rng.rand(200000, 2).astype(np.float32)
    wi = unit((0.3, 0.2, 0.9))
    wo, w, pdf, e, delta = bsdf_sample(oracle_lib, mat, wi, xy)
    null = np.all(np.isclose(wo, -wi, atol=1e-7), axis=1)
    assert abs(null.mean() - (1 - prob)) < 0.004 and np.all(delta[null] == 1) and np.all(delta[~null] == 0)
    assert np.allclose(pdf[null], 1 - prob, rtol=1e-6) and np.allclose(w[null], (1 - op) / (1 - prob), rtol=1e-5)
    f, p = bsdf_eval(oracle_lib, mat, wi, wo[~null])
    f0, p0 = bsdf_eval(oracle_lib, base, wi, wo[~null])
    assert np.allclose(f, f0 * op, rtol=1e-6) and np.allclose(p, p0 * prob, rtol=1e-6)
    assert np.allclose(w[~null], f / p[:, None], rtol=2e-3) and np.allclose(pdf[~null], p, rtol=2e-3)
    # energy: E[weight] = opacity * albedo + (1 - opacity)
    assert np.allclose(w.astype(np.float64).mean(0), op * np.float32([0.5, 0.6, 0.7]) + (1 - op), atol=0.01)
    m = Material.from_dict(mat)
    a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
    oracle_lib.ppgo_bsdf_flags(C.byref(m), C.byref(a), C.byref(b), C.byref(c))
    assert (a.value, b.value, c.value) == (1, 0, 1)  # still smooth (guided), not all-delta, EBackSide set by the null component
