"""Bitmap textures and bump maps (ppg_scene.texcoords / textures, ppg_material.texture).

Li fetches the BSDF with its.getBSDF() (GP:1934), so no UV partials exist and every bitmap lookup is BitmapTexture::eval(uv) =
MIPMap::evalBilinear(0, uv) (bitmap.cpp:431-452, mipmap.h:575-596).  Pins of the oracle: a floor under a uniform sky shows exactly
texture(uv) x L (the furnace identity, pixel by pixel against a numpy restatement of the bilinear lookup); the wrap modes; a bump map
of constant slope tilts the shading normal by the analytic angle.  GPU = oracle bit for bit on a scene using all of it.
"""
import ctypes as C
import os

import numpy as np
import pytest

import ppg_host
from conftest import CBOX_PROPS, IMPROVED, make_oracle


def _checker(h=8, w=16, seed=5):
    rng = np.random.RandomState(seed)
    return (0.1 + 0.8 * rng.rand(h, w, 3)).astype(np.float32)


def _floor_scene(res, tex, uv_scale=(1.0, 1.0), uv_offset=(0.0, 0.0), wrap=("repeat", "repeat"), nearest=False, uv_range=1.0, bump=None):
    """A 2 x 2 floor seen from straight above under a constant sky of radiance 2; texture coordinates (x, z) / 2 * uv_range."""
    pos = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    uv = ((pos[:, [0, 2]] + 1) / 2 * uv_range).astype(np.float32)
    idx = np.array([[0, 2, 1], [0, 3, 2]], np.uint32)  # normal +y
    mat = dict(type=0, reflectance=tuple(float(v) for v in tex.reshape(-1, 3).mean(0)), texture=0)
    textures = [dict(rgb=tex, uv_scale=uv_scale, uv_offset=uv_offset, wrap_u=wrap[0], wrap_v=wrap[1], nearest=nearest)]
    if bump is not None:
        mat["bump"] = 1
        textures.append(dict(rgb=bump, wrap_u="clamp", wrap_v="clamp"))
    cam = ppg_host.scenes.perspective_camera((0, 3, 0), (0, 0, 0), (0, 0, -1), 30.0, "x", 0.01, 100.0, res, res)
    return ppg_host.SceneDesc(pos, idx, np.zeros(2, np.uint32), np.full(2, -1, np.int32), [mat], [], cam, environment=(2.0, 2.0, 2.0), texcoords=uv, textures=textures)


def _np_lookup(tex, uv, scale, offset, wrap, nearest):
    """MIPMap::evalBilinear(0, uv) / evalBox with the boundary conditions of evalTexel (mipmap.h:503-596), float32 like the reference."""
    h, w, _ = tex.shape
    f = np.float32
    u = (uv[..., 0].astype(f) * f(scale[0]) + f(offset[0])).astype(f)
    v = (uv[..., 1].astype(f) * f(scale[1]) + f(offset[1])).astype(f)

    def texel(x, y):
        ok = np.ones(x.shape, bool)
        const = np.zeros(x.shape + (3,), f)
        out = []
        for c, size, mode in ((y, h, wrap[1]), (x, w, wrap[0])):  # evalTexel tests x first: its constant wins where both are outside
            if mode == "repeat":
                c = np.mod(c, size)
            elif mode == "clamp":
                c = np.clip(c, 0, size - 1)
            elif mode == "mirror":
                c = np.mod(c, 2 * size); c = np.where(c >= size, 2 * size - c - 1, c)
            else:
                outside = (c < 0) | (c >= size)
                ok &= ~outside
                const[outside] = 0.0 if mode == "zero" else 1.0
                c = np.clip(c, 0, size - 1)
            out.append(c)
        return np.where(ok[..., None], tex[out[0], out[1]], const)
    if nearest:
        return texel(np.floor(u * f(w)).astype(int), np.floor(v * f(h)).astype(int))
    uu, vv = (u * f(w) - f(0.5)).astype(f), (v * f(h) - f(0.5)).astype(f)
    x0, y0 = np.floor(uu).astype(int), np.floor(vv).astype(int)
    dx1, dy1 = (uu - x0).astype(f), (vv - y0).astype(f)
    dx2, dy2 = (f(1) - dx1).astype(f), (f(1) - dy1).astype(f)
    return (texel(x0, y0) * dx2[..., None] * dy2[..., None] + texel(x0, y0 + 1) * dx2[..., None] * dy1[..., None]
            + texel(x0 + 1, y0) * dx1[..., None] * dy2[..., None] + texel(x0 + 1, y0 + 1) * dx1[..., None] * dy1[..., None])


@pytest.mark.parametrize("kw", [dict(), dict(uv_scale=(3.0, 2.0), uv_offset=(0.25, -0.4)), dict(wrap=("mirror", "clamp"), uv_range=2.5, uv_offset=(-0.6, -0.7)),
                                dict(wrap=("zero", "one"), uv_range=1.6, uv_offset=(-0.3, -0.3)), dict(nearest=True, uv_scale=(2.0, 2.0))],
                         ids=["plain", "scale-offset", "mirror-clamp", "zero-one", "nearest"])
def test_textured_floor_under_a_uniform_sky_shows_the_texture(oracle_lib, kw):
    """Diffuse floor, constant sky L, one bounce: every sample returns reflectance(uv) x L exactly (cosine sampling: f cos / pdf = albedo), so
    the pixel mean over a pass IS the mean texture value over the pixel's jittered sample positions — compared against numpy's lookup at
    the same positions."""
    res, tex = 24, _checker()
    scene = _floor_scene(res, tex, **kw)
    e = make_oracle(oracle_lib, threads=2, budgetType="spp", budget=4, sppPerPass=4, maxDepth=2, rrDepth=10, seed=3)
    e.set_scene(scene); e.render()
    img = e.read_film()
    # the same camera samples, restated: pixel + (u1, u2) of the path's first two draws, ray through the pinhole onto y = 0
    lib = oracle_lib
    want = np.zeros((res, res, 3), np.float64)
    cam = scene.camera
    s2c, c2w = np.asarray(cam["sample_to_camera"], np.float64), np.asarray(cam["camera_to_world"], np.float64)
    for j in range(4):
        pix = np.arange(res * res)
        # ppg_path_key / ppg_rand through the oracle's math hook (op 4: rand(key, dim) with key = a, dim = b as bit patterns)
        px, py = pix % res, pix // res
        u = np.zeros((res * res, 2), np.float32)
        for d in range(2):
            a = np.zeros(res * res, np.float32); b = np.zeros(res * res, np.float32); o0 = np.zeros(res * res, np.float32); o1 = np.zeros(res * res, np.float32)
            a.view(np.uint32)[:] = pix
            b.view(np.uint32)[:] = (3 << 16) | (j << 4) | d  # seed 3, sample index, dimension (op 8)
            assert lib.ppgo_math_eval(8, res * res, a.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)),
                                      o0.ctypes.data_as(C.POINTER(C.c_float)), o1.ctypes.data_as(C.POINTER(C.c_float))) == 0
            u[:, d] = o0
        sx, sy = (px + u[:, 0]) / res, (py + u[:, 1]) / res
        near = (s2c @ np.stack([sx, sy, np.zeros_like(sx), np.ones_like(sx)])).T
        near = near[:, :3] / near[:, 3:4]
        dl = near / np.linalg.norm(near, axis=1, keepdims=True)
        dw = dl @ c2w[:3, :3].T
        o = c2w[:3, 3]
        t = -o[1] / dw[:, 1]
        hit = o[None] + dw * t[:, None]
        uv = ((hit[:, [0, 2]] + 1) / 2 * kw.get("uv_range", 1.0)).astype(np.float32)
        inside = (np.abs(hit[:, 0]) <= 1) & (np.abs(hit[:, 2]) <= 1)
        val = _np_lookup(tex, uv, kw.get("uv_scale", (1, 1)), kw.get("uv_offset", (0, 0)), kw.get("wrap", ("repeat", "repeat")), kw.get("nearest", False)) * 2.0
        val = np.where(inside[:, None], val, 2.0)  # rays that miss the floor see the sky itself
        want += val.reshape(res, res, 3) / 4
    # barycentric interpolation of uv and the hit point carry a few ulps; bilinear weights amplify them by the texture size
    assert np.abs(img - want).max() < 2e-3, np.abs(img - want).max()
    assert np.abs(img - want).mean() < 2e-5


def test_bump_map_of_constant_slope_tilts_the_shading_normal(oracle_lib):
    """Displacement h(u, v) = a u (a ramp): BumpMap::getFrame turns the shading normal by atan(a |du/dx|) about the v axis; under a sky
    that is bright on one side only the floor's radiance changes accordingly.  Checked through the render: a uniform sky gives exactly
    albedo x L whatever the tilt (cosine-weighted sampling in the perturbed frame), except for directions the adapter rejects
    (cosTheta(wo) cosTheta(wo') <= 0, bumpmap.cpp:170-171), whose share for a tilt t is (1 - cos t) / 2 of the cosine lobe."""
    res = 16
    ramp = np.repeat(np.linspace(0, 1, 64, dtype=np.float32)[None, :, None], 3, 2).repeat(4, 0) * 0.8  # dh/du = 0.8 (per unit u); du/dx = 0.5 → slope 0.4
    tex = np.full((2, 2, 3), 0.5, np.float32)
    scene = _floor_scene(res, tex, bump=ramp)
    e = make_oracle(oracle_lib, threads=2, budgetType="spp", budget=64, sppPerPass=4, maxDepth=2, rrDepth=10, seed=9)
    e.set_scene(scene); e.render()
    img = e.read_film()
    centre = img[4:12, 4:12].mean((0, 1))
    tilt = np.arctan(0.8 * 63 / 64 * 0.5)  # bilinear gradient between texel centres: (63 texel steps over 64 texels) x du/dx
    lost = (1 - np.cos(tilt)) / 2
    assert abs(centre[0] / (0.5 * 2.0 * (1 - lost)) - 1) < 0.01, (centre, lost)


def test_flat_scene_file_round_trips_textures(tmp_path):
    tex = _checker()
    scene = _floor_scene(8, tex, uv_scale=(2.0, 1.0), wrap=("mirror", "repeat"), bump=_checker(4, 4, 2))
    scene.textures[0]["srgb8"] = (np.random.RandomState(1).rand(8, 16, 3) * 255).astype(np.uint8)
    scene.textures[0]["rgb"] = ppg_host.scenes.srgb8_table()[scene.textures[0]["srgb8"]]
    p = str(tmp_path / "s.ppgs")
    ppg_host.save_scene(scene, p)
    back = ppg_host.load_scene_file(p)
    assert np.array_equal(back.texcoords, scene.texcoords) and len(back.textures) == 2
    for a, b in zip(back.textures, scene.textures):
        assert np.array_equal(a["rgb"], b["rgb"]) and a["wrap_u"] == b.get("wrap_u", "repeat") and tuple(a["uv_scale"]) == tuple(float(v) for v in b.get("uv_scale", (1, 1)))
    assert back.materials[0]["texture"] == 0 and back.materials[0]["bump"] == 1


def _textured_cbox(res):
    """CBOX with a bitmap on the floor (two-sided diffuse), a textured rough-plastic tall box with a bump map, texture coordinates on both."""
    scene = ppg_host.cbox_scene(*res)
    pos = np.asarray(scene.positions, np.float32)
    uv = np.full((pos.shape[0], 2), np.nan, np.float32)
    idx = np.asarray(scene.indices)
    tm = np.asarray(scene.tri_material).copy()
    floor_tris = np.where(np.all(np.abs(pos[idx][:, :, 1]) < 1e-3, axis=1))[0]
    box_tris = np.arange(24, 36)  # tall box
    for t in np.concatenate([floor_tris, box_tris]):
        for v in idx[t]:
            uv[v] = (pos[v, 0] / 556.0 * 3.0, (pos[v, 2] + pos[v, 1]) / 556.0 * 2.0)
    rt = None
    mats = list(scene.materials)
    mats.append(dict(type=1, reflectance=(0.5, 0.5, 0.5), texture=0))
    mats.append(dict(type=5, reflectance=(0.4, 0.4, 0.4), specular=(1.0, 1.0, 1.0), eta=1.49, texture=0, bump=1, twosided=True))
    tm[floor_tris] = len(mats) - 2
    tm[box_tris] = len(mats) - 1
    scene.materials = mats
    scene.tri_material = tm
    scene.texcoords = uv
    rng = np.random.RandomState(4)
    scene.textures = [dict(rgb=(0.05 + 0.9 * rng.rand(16, 16, 3)).astype(np.float32)),
                      dict(rgb=np.repeat(rng.rand(32, 32, 1).astype(np.float32) * 0.05, 3, 2), wrap_u="mirror", wrap_v="clamp", uv_scale=(2.0, 1.5))]
    return scene


def test_textured_cbox_renders_and_differs_from_the_untextured_one(oracle_lib):
    props = dict(CBOX_PROPS, budget=12, seed=6)
    e = make_oracle(oracle_lib, threads=4, **props)
    e.set_scene(_textured_cbox((48, 48))); e.render()
    a = e.read_film()
    e2 = make_oracle(oracle_lib, threads=4, **props)
    e2.set_scene(ppg_host.cbox_scene(48, 48)); e2.render()
    b = e2.read_film()
    assert np.isfinite(a).all() and a.mean() > 0.01 and np.abs(a - b).mean() > 0.005


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [{}, IMPROVED, dict(nee="kickstart", maxDepth=-1, rrDepth=4, strictNormals=0)], ids=["default", "improved", "kickstart-unbounded"])
def test_textures_and_bump_maps_against_oracle(oracle_lib, extra):
    from test_gpu_parity import assert_tree_equal, hip
    scene = _textured_cbox((64, 48))
    props = dict(CBOX_PROPS, budget=28 if not extra else 31, seed=12, **extra)
    g = hip(**props)
    ig = ppg_host.GuidedPathTracer(engine=g).render(scene)
    o = make_oracle(oracle_lib, threads=os.cpu_count() or 8, **props)
    io = ppg_host.GuidedPathTracer(engine=o).render(scene)
    assert np.isfinite(ig).all() or np.array_equal(np.isfinite(ig), np.isfinite(io))
    assert np.array_equal(ig, io, equal_nan=True)
    assert_tree_equal(g.read_sdtree(), o.read_sdtree())


@pytest.mark.gpu
def test_textured_floor_on_the_gpu_equals_the_oracle(oracle_lib):
    from test_gpu_parity import hip
    for kw in (dict(uv_scale=(3.0, 2.0), uv_offset=(0.25, -0.4)), dict(wrap=("mirror", "clamp"), uv_range=2.5, uv_offset=(-0.6, -0.7)), dict(wrap=("zero", "one"), uv_range=1.6, uv_offset=(-0.3, -0.3)),
               dict(nearest=True, uv_scale=(2.0, 2.0)), dict(bump=_checker(8, 8, 3))):
        scene = _floor_scene(32, _checker(), **kw)
        props = dict(budgetType="spp", budget=8, sppPerPass=4, maxDepth=3, rrDepth=10, seed=3)
        g = hip(**props); g.set_scene(scene); g.render()
        o = make_oracle(oracle_lib, threads=4, **props); o.set_scene(scene); o.render()
        assert np.array_equal(g.read_film(), o.read_film()), kw


def _sphere_only_scene(res):
    """No triangle at all: one diffuse analytic sphere under a uniform sky (a convex body: every reflected ray leaves the scene)."""
    cam = ppg_host.scenes.perspective_camera((0, 0, 4), (0, 0, 0), (0, 1, 0), 30.0, "x", 0.01, 100.0, res, res)
    return ppg_host.SceneDesc(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.int32),
                              [dict(type=0, reflectance=(0.25, 0.5, 0.75))], [], cam, environment=(2.0, 2.0, 2.0),
                              spheres=[dict(center=(0, 0, 0), radius=1.0, material=0, emitter=-1)])


def test_scene_of_analytic_spheres_only(oracle_lib):
    """A scene Mitsuba renders must not be refused because it holds no triangle mesh: the furnace identity on a convex diffuse sphere."""
    e = make_oracle(oracle_lib, threads=2, budgetType="spp", budget=8, sppPerPass=4, maxDepth=4, rrDepth=10, seed=2)
    e.set_scene(_sphere_only_scene(24)); e.render()
    img = e.read_film()
    assert np.allclose(img[10:14, 10:14], np.array([0.25, 0.5, 0.75]) * 2.0, rtol=1e-5)   # on the sphere: albedo x L, sample by sample
    assert np.allclose(img[0, 0], 2.0)                                                      # beside it: the sky


@pytest.mark.gpu
def test_scene_of_analytic_spheres_only_on_the_gpu(oracle_lib):
    from test_gpu_parity import hip
    props = dict(budgetType="spp", budget=28, sppPerPass=4, maxDepth=6, rrDepth=3, seed=2)
    scene = _sphere_only_scene(48)
    g = hip(**props); g.set_scene(scene); g.render()
    o = make_oracle(oracle_lib, threads=4, **props); o.set_scene(scene); o.render()
    assert np.array_equal(g.read_film(), o.read_film())
