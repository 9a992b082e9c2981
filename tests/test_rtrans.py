"""roughplastic's rough-transmittance slices (include/ppg.h: ppg_scene.rtrans; ppg_host/rtrans.py, host/rough_transmittance.h).

The committed fixture tests/golden/rtrans_slices.npz was cut from Mitsuba's data/microfacet tables by make_rtrans_slices.py.  Pins:
  * physics: the reference's table is the albedo of the rough dielectric TRANSMISSION lobe — integrating the oracle's own
    roughdielectric lobe reproduces the slices to 3-4 digits, which checks the file parser, the reduction (setEta / setAlpha), the
    interpolation AND the oracle's roughdielectric transmission term against numbers the reference shipped;
  * where the reference tree is mounted (development container): regenerating the slices reproduces the fixture bit for bit.
"""
import ctypes as C
import os

import numpy as np
import pytest

import ppg_host
from ppg_host import rtrans
from test_bsdfs import bsdf_eval, sphere_grid

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "rtrans_slices.npz")
REF_DATA = "/root/reference/mitsuba/data"
needs_ref = pytest.mark.skipif(not os.path.exists(os.path.join(REF_DATA, "microfacet", "ggx.dat")), reason="Mitsuba data tables not mounted")


def cases():
    g = np.load(GOLDEN)
    return [((str(d), float(a), float(e)), g["slice%d" % i]) for i, (d, a, e) in enumerate(g["cases"])]


def test_slices_are_the_transmission_albedo_of_the_rough_dielectric_lobe(oracle_lib):
    d, dw, _ = sphere_grid()
    below = d[:, 2] < 0
    for (distr, alpha, eta), sl in cases():
        assert sl.shape == (101,) and np.all((sl >= 0) & (sl <= 1))
        mat = dict(type="roughdielectric", alpha=alpha, eta=eta, reflectance=(0, 0, 0), specular=(1, 1, 1), distribution=distr)
        for mu in (1.0, 0.7, 0.3, 0.1):
            wi = np.float32([np.sqrt(1 - mu * mu), 0, mu])
            f, _ = bsdf_eval(oracle_lib, mat, wi, d)
            # eval() carries the radiance scaling 1/eta^2 of a refracted ray (roughdielectric.cpp:336-339); energy does not
            albedo = float((f[:, 0].astype(np.float64) * dw)[below].sum()) * eta * eta
            T = float(rtrans.cubic_interp_1d(np.float32(mu) ** np.float32(0.25), sl[:-1]))
            assert abs(T - albedo) < 1.5e-3, (distr, alpha, eta, mu, T, albedo)
        # last entry: diffuse transmittance from the inside = cosine-weighted hemispherical average of the transmission albedo with
        # the relative IOR inverted (rtrans.h:249-258 after setEta(1 / eta))
        inside = dict(mat, eta=float(np.float32(1) / np.float32(eta)))
        mus = (np.arange(24) + 0.5) / 24
        Ts = []
        for mu in mus:
            wi = np.float32([np.sqrt(1 - mu * mu), 0, mu])
            f, _ = bsdf_eval(oracle_lib, inside, wi, d)
            Ts.append(float((f[:, 0].astype(np.float64) * dw)[below].sum()) / (eta * eta))
        diffuse = float(np.sum(2 * mus * np.array(Ts)) / 24)
        assert abs(diffuse - float(sl[-1])) < 6e-3, (distr, alpha, eta, diffuse, sl[-1])


@needs_ref
def test_reduction_of_the_reference_tables_reproduces_the_fixture():
    g = np.load(GOLDEN)
    for (distr, alpha, eta), sl in cases():
        assert np.array_equal(rtrans.roughplastic_slice(distr, alpha, eta, REF_DATA), sl)
    for d in ("ggx", "beckmann"):
        t = rtrans.RoughTransmittance(os.path.join(REF_DATA, "microfacet", d + ".dat"))
        assert list(g[d + "_shape"]) == [t.eta_samples, t.alpha_samples, t.theta_samples] == [50, 50, 100]
        assert np.array_equal(g[d + "_probe"], np.concatenate([t.trans[7, 11, ::9], t.trans[53, 40, ::9], t.diff[::17, 5]]))
    with pytest.raises(rtrans.RoughTransmittanceError, match="outside the tabulated range"):
        rtrans.roughplastic_slice("ggx", 0.1, 5.0, REF_DATA)
    with pytest.raises(rtrans.RoughTransmittanceError, match="alpha"):
        rtrans.roughplastic_slice("ggx", 4.5, 1.5, REF_DATA)


def test_interpolation_restatement():
    """cubic_interp_1d / cubic_interp_nd: exact at the knots, reproduce cubics' neighbours smoothly, and agree with each other on
    separable data (the 2D / 3D forms are tensor products of the 1D weights, spline.cpp:236-452)."""
    rng = np.random.RandomState(2)
    v = rng.rand(9).astype(np.float32)
    for k in range(9):
        assert rtrans.cubic_interp_1d(np.float32(k) / np.float32(8), v) == v[k]
    assert rtrans.cubic_interp_1d(np.float32(1.0000001), v) == 0 and rtrans.cubic_interp_1d(np.float32(np.nan), v) == 0
    a, b, c = rng.rand(7).astype(np.float32), rng.rand(5).astype(np.float32), rng.rand(6).astype(np.float32)
    grid3 = (c[:, None, None] * b[None, :, None] * a[None, None, :]).astype(np.float32)      # [z][y][x]
    for x, y, z in rng.rand(20, 3).astype(np.float32):
        sep = float(rtrans.cubic_interp_1d(x, a)) * float(rtrans.cubic_interp_1d(y, b)) * float(rtrans.cubic_interp_1d(z, c))
        assert abs(float(rtrans.cubic_interp_nd([x, y, z], grid3)) - sep) < 2e-6
        sep2 = float(rtrans.cubic_interp_1d(x, a)) * float(rtrans.cubic_interp_1d(y, b)) * float(c[2])
        assert abs(float(rtrans.cubic_interp_nd([x, y], grid3[2])) - sep2) < 2e-6


def test_loader_needs_the_mitsuba_data_directory(tmp_path, monkeypatch):
    from test_mitsuba_xml import _write
    monkeypatch.delenv("PPG_MITSUBA_DATA", raising=False)
    xml = _write(tmp_path, '<shape type="rectangle"><bsdf type="roughplastic"/></shape>')
    with pytest.raises(ppg_host.mitsuba_xml.SceneError, match="data/microfacet"):
        ppg_host.load_scene(xml, defines=dict(nee="never"))
    with pytest.raises(ppg_host.mitsuba_xml.SceneError, match="not found"):
        ppg_host.load_scene(xml, defines=dict(nee="never"), data_dir=str(tmp_path))


@needs_ref
def test_roughplastic_in_scene_xml_and_flat_file(tmp_path):
    from test_mitsuba_xml import _write
    from ppg_host.bindings import Material
    xml = _write(tmp_path, """
    <shape type="rectangle"><bsdf type="roughplastic"><string name="distribution" value="ggx"/><float name="alpha" value="0.1"/>
        <float name="intIOR" value="1.5"/><float name="extIOR" value="1"/><rgb name="diffuseReflectance" value="0.1, 0.2, 0.3"/></bsdf></shape>
    <shape type="rectangle"><bsdf type="twosided"><bsdf type="roughplastic"><float name="alpha" value="0.2"/><boolean name="nonlinear" value="true"/></bsdf></bsdf></shape>
    <shape type="rectangle"><bsdf type="roughplastic"><string name="distribution" value="ggx"/><float name="alpha" value="0.1"/>
        <float name="intIOR" value="1.5"/><float name="extIOR" value="1"/><rgb name="diffuseReflectance" value="0.5, 0.5, 0.5"/></bsdf></shape>""")
    desc, _, info = ppg_host.load_scene(xml, defines=dict(nee="never"), data_dir=REF_DATA)
    rp = [m for m in desc.materials if m["type"] == 9]
    assert len(rp) == 3 and desc.rtrans.shape == (2, 101)                  # two distinct (distribution, alpha, eta): slices are shared
    assert [m["rtrans"] for m in rp] == [0, 1, 0] and rp[1]["twosided"] and rp[1]["nonlinear"] and rp[1]["distribution"] == "beckmann"
    g = dict(cases())
    assert np.array_equal(desc.rtrans[0], g[("ggx", 0.1, 1.5)])
    assert np.array_equal(desc.rtrans[1], rtrans.roughplastic_slice("beckmann", 0.2, np.float32(1.49 / 1.000277), REF_DATA))   # polypropylene / air
    assert Material.from_dict(rp[1]).rtrans == 1
    # flat file and XML round trips keep the slices
    p = str(tmp_path / "s.ppgs")
    ppg_host.save_scene(desc, p)
    raw = open(p, "rb").read()
    tail = np.frombuffer(raw[-(8 + 4 * 202):], np.uint8)
    assert list(np.frombuffer(tail[:8].tobytes(), np.uint32)) == [2, 100] and np.array_equal(np.frombuffer(tail[8:].tobytes(), np.float32).reshape(2, 101), desc.rtrans)
    back, _, _ = ppg_host.load_scene(ppg_host.save_scene_xml(desc, dict(budgetType="spp", budget=8.0), str(tmp_path / "rt")), data_dir=REF_DATA)
    assert np.array_equal(np.sort(back.rtrans, 0), np.sort(desc.rtrans, 0))
