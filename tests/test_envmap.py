"""Image-based environment emitter (emitters/envmap.cpp → ppg_scene.envmap): the oracle's restatement element-wise, the way the
reference validates sampling code (chi-square style), analytic renders, and the HDR image readers of the loaders."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import ppg_host
from ppg_host import imageio
from ppg_host.bindings import EnvMap, _fp


def _sun_map(h=32, w=64, seed=1):
    rng = np.random.RandomState(seed)
    rgb = (rng.rand(h, w, 3) ** 4).astype(np.float32) * 2
    rgb[6:9, 40:44] += 60.0           # a "sun" high in the sky
    return rgb


def _sphere_grid(n_t=720, n_p=1440):
    th = (np.arange(n_t) + 0.5) / n_t * np.pi
    ph = (np.arange(n_p) + 0.5) / n_p * 2 * np.pi
    TH, PH = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(TH) * np.sin(PH), np.cos(TH), -np.sin(TH) * np.cos(PH)], -1).reshape(-1, 3).astype(np.float32)  # envmap.cpp:590-593
    return d, (np.sin(TH) * (np.pi / n_t) * (2 * np.pi / n_p)).reshape(-1)


def test_sampling_density_and_weights(oracle_lib):
    a = np.float32(0.7)
    rot = [np.cos(a), 0, np.sin(a), 0, 1, 0, -np.sin(a), 0, np.cos(a)]
    for to_world in (np.eye(3).reshape(-1), rot):
        em = EnvMap.from_dict(dict(rgb=_sun_map(), scale=1.5, to_world=to_world))
        d, dw = _sphere_grid()
        val = np.zeros_like(d); pdf = np.zeros(len(d), np.float32)
        assert oracle_lib.ppgo_envmap_eval(C.byref(em), len(d), _fp(d), _fp(val), _fp(pdf)) == 0
        assert abs(float((pdf.astype(np.float64) * dw).sum()) - 1) < 1e-3           # a density over the sphere
        power = (val.astype(np.float64) * dw[:, None]).sum(0)
        xy = np.random.RandomState(3).rand(400000, 2).astype(np.float32)
        do = np.zeros((len(xy), 3), np.float32); wgt = np.zeros((len(xy), 3), np.float32); p2 = np.zeros(len(xy), np.float32)
        assert oracle_lib.ppgo_envmap_sample(C.byref(em), len(xy), _fp(xy), _fp(do), _fp(wgt), _fp(p2)) == 0
        assert np.abs(np.linalg.norm(do, axis=1) - 1).max() < 1e-6
        assert np.allclose(wgt.astype(np.float64).mean(0), power, rtol=0.01)        # E[value / pdf] = the map's integral
        v2 = np.zeros_like(do); p3 = np.zeros(len(do), np.float32)
        oracle_lib.ppgo_envmap_eval(C.byref(em), len(do), _fp(do), _fp(v2), _fp(p3))
        ok = p2 > 0
        assert np.percentile(np.abs(p3[ok] / p2[ok] - 1), 99) < 1e-3               # pdfDirect(sampled direction) = the sampling density
        assert np.percentile(np.abs(v2[ok] - wgt[ok] * p2[ok, None]).max(1) / (np.abs(v2[ok]).max(1) + 1e-6), 99) < 1e-3
        # histogram of the sampled directions against the integrated density, 12 x 16 bins in (theta, phi)
        n_t, n_p = 720, 1440
        H = (pdf.astype(np.float64) * dw).reshape(12, n_t // 12, 16, n_p // 16).sum((1, 3))
        dd, _ = _sphere_grid()
        loc = do.astype(np.float64) @ np.reshape(to_world, (3, 3)).astype(np.float64)   # world → emitter frame (transpose of a rotation)
        dl = dd.astype(np.float64) @ np.reshape(to_world, (3, 3)).astype(np.float64)
        def bins(v):
            it = np.clip((np.arccos(np.clip(v[:, 1], -1, 1)) / np.pi * 12).astype(int), 0, 11)
            ip = np.clip(((np.arctan2(v[:, 0], -v[:, 2]) % (2 * np.pi)) / (2 * np.pi) * 16).astype(int), 0, 15)
            return it, ip
        Hl = np.zeros((12, 16)); np.add.at(Hl, bins(dl), pdf.astype(np.float64) * dw)
        cnt = np.zeros((12, 16)); np.add.at(cnt, bins(loc), 1.0)
        big = Hl > 2e-3
        assert big.sum() > 20 and np.allclose(cnt[big] / len(xy), Hl[big], rtol=0.08, atol=3e-4)
        assert H.sum() == pytest.approx(1, abs=1e-3)
    black = EnvMap.from_dict(dict(rgb=np.zeros((4, 8, 3), np.float32)))
    assert oracle_lib.ppgo_envmap_eval(C.byref(black), 0, None, None, None) != 0    # "completely black -- this is not allowed"


def _floor(res):
    from test_oracle_known_answers import _floor_and_lamp
    scene = _floor_and_lamp(res)
    scene.indices, scene.tri_material, scene.tri_emitter = scene.indices[:2], scene.tri_material[:2], scene.tri_emitter[:2]
    scene.emitters = []
    return scene


def test_floor_under_an_environment_map(oracle_lib):
    """Direct light only (maxDepth 2) on a diffuse floor of albedo 0.5: radiance = albedo / pi x the cosine-weighted integral of the map
    over the upper hemisphere — for BSDF sampling (escaped rays look the map up, GP:2236-2243, envmap.cpp:381-407) and for luminaire
    sampling (envmap.cpp:510-538) combined by MIS with EnvironmentMap::pdfDirect.  A constant map must reproduce the `constant`
    emitter's albedo x L; a rotated map must move the sun with it."""
    from conftest import make_oracle
    res = 8
    d, dw = _sphere_grid(360, 720)
    up = d[:, 1] > 0
    for name, rgb, scale, rot in (("constant", np.full((8, 16, 3), 2.0, np.float32) * np.float32([1, 2, 3]), 0.5, None), ("sun", _sun_map(), 1.5, None),
                                  ("sun-rotated", _sun_map(), 1.0, 1.1)):
        R = np.eye(3, dtype=np.float32)
        if rot is not None:  # about z: tips the sun towards / away from the floor normal
            R = np.array([[np.cos(rot), -np.sin(rot), 0], [np.sin(rot), np.cos(rot), 0], [0, 0, 1]], np.float32)
        em = EnvMap.from_dict(dict(rgb=rgb, scale=scale, to_world=R.reshape(-1)))
        val = np.zeros_like(d); pdf = np.zeros(len(d), np.float32)
        oracle_lib.ppgo_envmap_eval(C.byref(em), len(d), _fp(d), _fp(val), _fp(pdf))
        expect = 0.5 / np.pi * (val[up].astype(np.float64) * (d[up, 1].astype(np.float64) * dw[up])[:, None]).sum(0)
        if name == "constant":
            assert np.allclose(expect, 0.5 * np.float32([1, 2, 3]), rtol=2e-3)
        scene = _floor(res)
        scene.envmap = dict(rgb=rgb, scale=scale, to_world=R.reshape(-1))
        for nee, budget, tol in (("always", 256, 0.02), ("never", 4096, 0.06 if name != "constant" else 0.005)):
            e = make_oracle(oracle_lib, threads=16, budgetType="spp", budget=budget, maxDepth=2, rrDepth=10, nee=nee, seed=9)
            img = ppg_host.GuidedPathTracer(engine=e).render(scene)
            got = img.reshape(-1, 3).astype(np.float64).mean(0)
            assert np.allclose(got, expect, rtol=tol), (name, nee, got, expect)
    both = _floor(4)
    both.envmap = dict(rgb=_sun_map()); both.environment = (1, 1, 1)
    with pytest.raises(ppg_host.PPGError, match="one environment emitter"):
        ppg_host.GuidedPathTracer(engine=make_oracle(oracle_lib, budgetType="spp", budget=4)).render(both)


def _write_hdr(path, img, rle):
    """Radiance RGBE writer for the test (shared exponent = that of the largest channel, 8-bit mantissas)."""
    H, W, _ = img.shape
    m = img.max(2)
    e = np.where(m > 1e-32, np.floor(np.log2(np.maximum(m, 1e-38))).astype(int) + 1, -128)
    scale = np.where(m > 1e-32, np.ldexp(256.0, -e), 0.0)
    rgbe = np.zeros((H, W, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = np.where(m > 1e-32, e + 128, 0)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (H, W))
        for y in range(H):
            if not rle:
                f.write(rgbe[y].tobytes()); continue
            f.write(bytes([2, 2, W >> 8, W & 255]))
            for ch in range(4):
                row, x = rgbe[y, :, ch], 0
                while x < W:
                    run = 1
                    while x + run < W and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 4:
                        f.write(bytes([128 + run, row[x]])); x += run
                    else:
                        n = min(W - x, 100)
                        f.write(bytes([n]) + row[x:x + n].tobytes()); x += n
    f = np.where(rgbe[..., 3] > 0, np.ldexp(1.0, rgbe[..., 3].astype(int) - 136), 0.0)
    return (rgbe[..., :3] * f[..., None]).astype(np.float32)


def test_hdr_image_readers_and_the_envmap_element(tmp_path):
    from test_mitsuba_xml import _write
    rng = np.random.RandomState(4)
    img = (rng.rand(6, 40, 3) ** 3 * 50).astype(np.float32)
    img[2, 5:30] = img[2, 5]                      # a run, so that the RLE path has something to do
    for rle in (False, True):
        p = str(tmp_path / ("m%d.hdr" % rle))
        q = _write_hdr(p, img, rle)
        assert np.array_equal(imageio.read_hdr(p), q) and np.all(np.abs(q - img) <= img.max(2, keepdims=True) / 128)   # 8-bit mantissas under a shared exponent
    imageio.write_pfm(str(tmp_path / "m.pfm"), img)
    imageio.write_exr(str(tmp_path / "m.exr"), img)
    assert np.array_equal(imageio.read_image(str(tmp_path / "m.pfm")), img) and np.array_equal(imageio.read_image(str(tmp_path / "m.exr")), img)
    xml = _write(tmp_path, '<emitter type="envmap"><string name="filename" value="m.pfm"/><float name="scale" value="2.5"/>'
                           '<transform name="toWorld"><rotate y="1" angle="90"/></transform></emitter>')
    desc, _, _ = ppg_host.load_scene(xml, defines=dict(nee="never"))
    assert np.array_equal(desc.envmap["rgb"], img) and desc.envmap["scale"] == 2.5
    assert np.allclose(np.reshape(desc.envmap["to_world"], (3, 3)), [[0, 0, 1], [0, 1, 0], [-1, 0, 0]], atol=1e-6)
    p = str(tmp_path / "s.ppgs")
    ppg_host.save_scene(desc, p)
    back = ppg_host.load_scene_file(p)
    assert np.array_equal(back.envmap["rgb"], img) and back.envmap["scale"] == 2.5 and struct.unpack_from("<6I", open(p, "rb").read(), 4)[5] & 8
    rt, _, _ = ppg_host.load_scene(ppg_host.save_scene_xml(desc, dict(budgetType="spp", budget=8.0), str(tmp_path / "rt")))
    assert np.array_equal(rt.envmap["rgb"], img) and np.allclose(rt.envmap["to_world"], desc.envmap["to_world"], atol=1e-6)
    for bad, needle in (('<emitter type="envmap"><string name="filename" value="nope.exr"/></emitter>', "not found"),
                        ('<emitter type="envmap"><string name="filename" value="m.pfm"/><transform name="toWorld"><scale value="2"/></transform></emitter>', "rotation"),
                        ('<emitter type="envmap"><string name="filename" value="m.pfm"/></emitter><emitter type="constant"/>', "not supported")):
        with pytest.raises(ppg_host.mitsuba_xml.SceneError, match=needle):
            ppg_host.load_scene(_write(tmp_path, bad), defines=dict(nee="never"))
