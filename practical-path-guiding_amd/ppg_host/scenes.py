"""Procedural scene descriptions handed to ppg_set_scene().

`cbox_scene()` restates /root/reference/scenes/cbox/cbox.xml + meshes/*.obj: 36 triangles, five
Lambertian BSDFs, the upside-down area light, the pinhole camera.  Coordinates are the vertex data
of the classic Cornell box; RGB values were derived from the XML's spectra with
tools/derive_cbox_rgb.py (dev-time; it reads the CIE tables from the reference checkout).  The GPU
box has no /root/reference, hence code instead of files.

`room_scene()` is the "kitchen-class" stand-in of SURVEY.md §8(d) S3: a closed room lit only through
a slit, filled with tessellated boxes — many triangles, hard indirect light; Lambertian, or (glossy=True) the S3 material mix
with GGX metal and plastic.
"""
import math

import numpy as np


class SceneDesc:
    def __init__(self, positions, indices, tri_material, tri_emitter, materials, emitters, camera, normals=None, environment=None, rtrans=None, spheres=None, envmap=None,
                 texcoords=None, textures=None):
        self.positions, self.indices = positions, indices
        self.tri_material, self.tri_emitter = tri_material, tri_emitter
        self.materials, self.emitters, self.camera, self.normals = materials, emitters, camera, normals
        self.environment = environment  # None or (r, g, b): constant environment emitter
        self.rtrans = rtrans            # None or float32 [n_slices, samples + 1]: rough-transmittance slices of the roughplastic materials
        self.envmap = envmap            # None or dict(rgb=float32 [h, w, 3], scale, to_world): image-based environment emitter (bindings.EnvMap)
        self.spheres = spheres or []    # analytic spheres: dicts {center, radius, material, emitter, flip_normals, to_world} (bindings.Sphere)
        self.texcoords = texcoords      # None or float32 [n_vertices, 2] (NaN rows: vertices of meshes without texture coordinates)
        self.textures = textures or []  # bitmap textures: dicts {rgb, uv_scale, uv_offset, wrap_u, wrap_v, nearest} (bindings.Texture); materials refer
                                        # to them by index: material["texture"] (diffuse reflectance), material["bump"] (bumpmap displacement)

    @property
    def n_triangles(self):
        return int(np.asarray(self.indices).shape[0])


def save_scene(desc, path):
    """Write the flat binary scene read by host/ppg_render.cpp: "PPGS", 6 x uint32 {n_vertices, n_triangles, n_materials,
    n_emitters, has_normals, blocks (bit 0: environment, bit 1: rtrans, bit 2: spheres, bit 3: envmap, bit 4: texcoords, bit 5: textures)}, then positions, [normals], indices, tri_material,
    tri_emitter, materials (ppg_material, 80 bytes each), emitters (4 floats), camera (ppg_camera), [environment radiance:
    3 floats], [rtrans: 2 x uint32 {n_slices, samples}, then n_slices x (samples + 1) floats], [spheres: uint32 n, then n x ppg_sphere
    (64 bytes)], [envmap: 2 x uint32 {width, height}, float scale, 9 floats to_world, then height x width x 3 floats], [texcoords: n_vertices x 2
    floats], [textures: uint32 n, then per texture 2 x uint32 {width, height}, 4 floats uv scale / offset, 3 x int32 {wrap_u, wrap_v, nearest}, uint32
    storage (0: float32 RGB, 1: uint8 sRGB-encoded RGB — decoded on load with the exact 256-entry table of the 8-bit → float conversion), pixels]."""
    import struct
    pos = np.ascontiguousarray(desc.positions, np.float32)
    idx = np.ascontiguousarray(desc.indices, np.uint32)
    with open(path, "wb") as f:
        f.write(b"PPGS")
        env = getattr(desc, "environment", None)
        rt = getattr(desc, "rtrans", None)
        rt = None if rt is None or not len(rt) else np.ascontiguousarray(rt, np.float32)
        f.write(struct.pack("<6I", pos.shape[0], idx.shape[0], len(desc.materials), len(desc.emitters), 0 if desc.normals is None else 1,
                            (0 if env is None else 1) | (0 if rt is None else 2) | (4 if getattr(desc, "spheres", None) else 0) | (8 if getattr(desc, "envmap", None) is not None else 0)
                            | (16 if getattr(desc, "texcoords", None) is not None else 0) | (32 if getattr(desc, "textures", None) else 0)))
        f.write(pos.tobytes())
        if desc.normals is not None:
            f.write(np.ascontiguousarray(desc.normals, np.float32).tobytes())
        f.write(idx.tobytes())
        f.write(np.ascontiguousarray(desc.tri_material, np.uint32).tobytes())
        f.write(np.ascontiguousarray(desc.tri_emitter, np.int32).tobytes())
        from .bindings import Material
        for m in desc.materials:
            f.write(bytes(Material.from_dict(m)))  # ppg_material, 80 bytes
        for e in desc.emitters:
            f.write(struct.pack("<4f", *[float(np.float32(v)) for v in e["radiance"]], 0))
        c = desc.camera
        f.write(np.asarray(c["sample_to_camera"], np.float32).tobytes())
        f.write(np.asarray(c["camera_to_world"], np.float32).tobytes())
        f.write(struct.pack("<2f2i", c["near_clip"], c["far_clip"], c["width"], c["height"]))
        if env is not None:
            f.write(struct.pack("<3f", *[float(np.float32(v)) for v in env]))
        if rt is not None:
            f.write(struct.pack("<2I", rt.shape[0], rt.shape[1] - 1))
            f.write(rt.tobytes())
        if getattr(desc, "spheres", None):
            from .bindings import Sphere
            f.write(struct.pack("<I", len(desc.spheres)))
            for d in desc.spheres:
                f.write(bytes(Sphere.from_dict(d)))
        if getattr(desc, "envmap", None) is not None:
            em = desc.envmap
            rgb = np.ascontiguousarray(em["rgb"], np.float32)
            f.write(struct.pack("<2If9f", rgb.shape[1], rgb.shape[0], float(np.float32(em.get("scale", 1.0))),
                                *[float(np.float32(v)) for v in np.asarray(em.get("to_world", np.eye(3)), np.float32).reshape(-1)]))
            f.write(rgb.tobytes())
        if getattr(desc, "texcoords", None) is not None:
            f.write(np.ascontiguousarray(desc.texcoords, np.float32).tobytes())
        if getattr(desc, "textures", None):
            from .bindings import Texture
            f.write(struct.pack("<I", len(desc.textures)))
            for t in desc.textures:
                rgb = np.ascontiguousarray(t["rgb"], np.float32)
                wu, wv = t.get("wrap_u", "repeat"), t.get("wrap_v", "repeat")
                src8 = t.get("srgb8")  # the 8-bit source of an sRGB-decoded image: stored instead of the floats (4x smaller)
                f.write(struct.pack("<2I4f3iI", rgb.shape[1], rgb.shape[0], *[float(np.float32(v)) for v in t.get("uv_scale", (1, 1))],
                                    *[float(np.float32(v)) for v in t.get("uv_offset", (0, 0))], Texture.WRAP[wu] if isinstance(wu, str) else int(wu),
                                    Texture.WRAP[wv] if isinstance(wv, str) else int(wv), 1 if t.get("nearest") else 0, 0 if src8 is None else 1))
                f.write(rgb.tobytes() if src8 is None else np.ascontiguousarray(src8, np.uint8).tobytes())


def srgb8_table():
    """uint8 sRGB → linear float32, the table of the 8-bit → float conversion (bitmap.cpp / fmtconv.cpp: value / 255 through the sRGB curve)."""
    v = np.arange(256, dtype=np.float64) / 255.0
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4).astype(np.float32)


def load_scene_file(path):
    """Read a flat scene written by save_scene() / `ppg_render --ppgs` back into a SceneDesc."""
    import struct
    from .bindings import Material, Sphere
    buf = open(path, "rb").read()
    if buf[:4] != b"PPGS":
        raise ValueError("%s: not a flat scene file" % path)
    nv, nt, nm, ne, has_n, blocks = struct.unpack_from("<6I", buf, 4)
    off = [28]

    def take(dtype, count):
        # every count comes from the file: checked against what is left of it (a corrupt or truncated file is a ValueError, not an
        # allocation of whatever size its header claims)
        need = int(count) * np.dtype(dtype).itemsize
        if count < 0 or need > len(buf) - off[0]:
            raise ValueError("%s: truncated or corrupt flat scene file" % path)
        a = np.frombuffer(buf, dtype, count, off[0])
        off[0] += a.nbytes
        return a

    def image_size():
        w, h = (int(v) for v in take(np.uint32, 2))
        if not (0 < w < 0x8000 and 0 < h < 0x8000):  # what ppg_set_scene accepts
            raise ValueError("%s: image of %d x %d pixels" % (path, w, h))
        return w, h
    pos = take(np.float32, 3 * nv).reshape(-1, 3).copy()
    nrm = take(np.float32, 3 * nv).reshape(-1, 3).copy() if has_n else None
    idx = take(np.uint32, 3 * nt).reshape(-1, 3).copy()
    tm, te = take(np.uint32, nt).copy(), take(np.int32, nt).copy()
    mats = []
    for _ in range(nm):
        m = Material.from_buffer_copy(bytes(take(np.uint8, 80)))
        d = dict(type=int(m.type), reflectance=tuple(m.reflectance), specular=tuple(m.specular), alpha=float(m.alpha), eta=tuple(m.eta), k=tuple(m.k),
                 twosided=bool(m.flags & 1), nonlinear=bool(m.flags & 2), rtrans=int(m.rtrans))
        if m.flags & 4:
            d["opacity"] = tuple(m.opacity)
        if m.flags & 8:
            d["distribution"] = "beckmann"
        if m.texture & 0xffff:
            d["texture"] = (m.texture & 0xffff) - 1
        if m.texture >> 16:
            d["bump"] = (m.texture >> 16) - 1
        mats.append(d)
    ems = [dict(radiance=tuple(float(v) for v in take(np.float32, 4)[:3])) for _ in range(ne)]
    cam = dict(sample_to_camera=take(np.float32, 16).reshape(4, 4).copy(), camera_to_world=take(np.float32, 16).reshape(4, 4).copy())
    cam["near_clip"], cam["far_clip"] = (float(v) for v in take(np.float32, 2))
    cam["width"], cam["height"] = (int(v) for v in take(np.int32, 2))
    env = tuple(float(v) for v in take(np.float32, 3)) if blocks & 1 else None
    rt = None
    if blocks & 2:
        n, samples = (int(v) for v in take(np.uint32, 2))
        rt = take(np.float32, n * (samples + 1)).reshape(n, samples + 1).copy()
    spheres = []
    if blocks & 4:
        for _ in range(int(take(np.uint32, 1)[0])):
            sp = Sphere.from_buffer_copy(bytes(take(np.uint8, 64)))
            spheres.append(dict(center=tuple(sp.center), radius=float(sp.radius), to_world=list(sp.to_world), material=int(sp.material), emitter=int(sp.emitter),
                                flip_normals=bool(sp.flip_normals)))
    envmap = None
    if blocks & 8:
        w, h = image_size()
        scale = float(take(np.float32, 1)[0])
        R = [float(v) for v in take(np.float32, 9)]
        envmap = dict(rgb=take(np.float32, w * h * 3).reshape(h, w, 3).copy(), scale=scale, to_world=R)
    uvs = take(np.float32, 2 * nv).reshape(-1, 2).copy() if blocks & 16 else None
    textures = []
    if blocks & 32:
        names = ["repeat", "mirror", "clamp", "zero", "one"]
        for _ in range(int(take(np.uint32, 1)[0])):
            w, h = image_size()
            sc = [float(v) for v in take(np.float32, 4)]
            wu, wv, nearest = (int(v) for v in take(np.int32, 3))
            storage = int(take(np.uint32, 1)[0])
            t = dict(uv_scale=tuple(sc[:2]), uv_offset=tuple(sc[2:]), wrap_u=names[wu], wrap_v=names[wv], nearest=bool(nearest))
            if storage == 1:
                t["srgb8"] = take(np.uint8, w * h * 3).reshape(h, w, 3).copy()
                t["rgb"] = srgb8_table()[t["srgb8"]]
            else:
                t["rgb"] = take(np.float32, w * h * 3).reshape(h, w, 3).copy()
            textures.append(t)
    if off[0] != len(buf):
        raise ValueError("%s: trailing bytes" % path)
    return SceneDesc(pos, idx, tm, te, mats, ems, cam, nrm, env, rt, spheres, envmap, uvs, textures)


def _sample_to_camera(fov_deg, fov_axis, near, far, width, height):
    """m_sampleToCamera of sensors/perspective.cpp:150-164; fov-axis handling of sensor.cpp:239-264.  Computed in
    double, stored as float32 (an input to both the HIP path and the oracle, not part of either)."""
    aspect = float(width) / float(height)
    axis = fov_axis
    if axis == "smaller":
        axis = "y" if aspect > 1 else "x"
    elif axis == "larger":
        axis = "x" if aspect > 1 else "y"
    if axis == "x":
        xfov = float(fov_deg)
    elif axis == "y":  # setYFov → xfov = 2 atan(tan(yfov / 2) * aspect)
        xfov = math.degrees(2.0 * math.atan(math.tan(0.5 * math.radians(fov_deg)) * aspect))
    elif axis == "diagonal":  # setDiagonalFov, sensor.cpp:296-301
        diagonal = 2.0 * math.tan(0.5 * math.radians(fov_deg))
        w = diagonal / math.sqrt(1.0 + 1.0 / (aspect * aspect))
        xfov = math.degrees(2.0 * math.atan(w * 0.5))
    else:
        raise ValueError("fovAxis %r not supported" % fov_axis)

    def scale(v):
        return np.diag([v[0], v[1], v[2], 1.0]).astype(np.float64)

    def translate(v):
        m = np.eye(4)
        m[:3, 3] = v
        return m

    recip = 1.0 / (far - near)
    cot = 1.0 / math.tan(math.radians(xfov / 2.0))
    persp = np.array([[cot, 0, 0, 0], [0, cot, 0, 0], [0, 0, far * recip, -near * far * recip], [0, 0, 1, 0]], np.float64)
    cam_to_sample = scale([-0.5, -0.5 * aspect, 1.0]) @ translate([-1.0, -1.0 / aspect, 0.0]) @ persp
    return np.linalg.inv(cam_to_sample).astype(np.float32)


def perspective_camera_from_matrix(camera_to_world, fov_deg, fov_axis, near, far, width, height):
    """Camera record from a ready toWorld matrix (what the XML loader has)."""
    return dict(sample_to_camera=_sample_to_camera(fov_deg, fov_axis, near, far, width, height),
                camera_to_world=np.asarray(camera_to_world, np.float32).reshape(4, 4), near_clip=float(near), far_clip=float(far),
                width=int(width), height=int(height), fov=float(fov_deg), fov_axis=str(fov_axis))


def resize_camera(camera, width, height):
    """The same perspective camera with another film size, keeping its HORIZONTAL field of view (fovAxis = x, Mitsuba's default and
    what the reference's bundled scenes use): the x-fov is read back from the projection (sample_to_camera^-1 [0][0] = -cot / 2)."""
    cam_to_sample = np.linalg.inv(np.asarray(camera["sample_to_camera"], np.float64).reshape(4, 4))
    xfov = math.degrees(2.0 * math.atan(1.0 / (-2.0 * cam_to_sample[0, 0])))
    return dict(camera, sample_to_camera=_sample_to_camera(xfov, "x", camera["near_clip"], camera["far_clip"], width, height),
                width=int(width), height=int(height))


def perspective_camera(origin, target, up, fov_deg, fov_axis, near, far, width, height):
    """Transform::lookAt (transform.cpp:191-214) + the projection above."""
    p, t, u = (np.asarray(v, np.float64) for v in (origin, target, up))
    d = (t - p) / np.linalg.norm(t - p)
    left = np.cross(u, d)
    left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = left, new_up, d, p
    return perspective_camera_from_matrix(c2w.astype(np.float32), fov_deg, fov_axis, near, far, width, height)


# linear RGB of the spectra in scenes/cbox/cbox.xml (tools/derive_cbox_rgb.py)
CBOX_RGB = {
    "box": (0.88579154, 0.698900044, 0.666440606),
    "white": (0.88579154, 0.698900044, 0.666440606),
    "red": (0.570083559, 0.0430142134, 0.0443643667),
    "green": (0.10540954, 0.377989352, 0.076434195),
    "light": (0.936401188, 0.740475953, 0.705280542),
}
CBOX_EMITTER_RGB = (36.7738686, 21.976778, 5.50738001)


def _extruded_box(footprint, h):
    """Top, four sides, bottom of a box whose top face has the (x, z) corners `footprint` in the
    winding of meshes/cbox_{small,large}box.obj (all faces wound to face outwards)."""
    a, b, c, d = footprint
    quads = [[(a[0], h, a[1]), (b[0], h, b[1]), (c[0], h, c[1]), (d[0], h, d[1])]]
    ring = [a, d, c, b]
    for i in range(4):
        p, q = ring[i], ring[(i + 1) % 4]
        quads.append([(p[0], 0.0, p[1]), (p[0], h, p[1]), (q[0], h, q[1]), (q[0], 0.0, q[1])])
    quads.append([(d[0], 0.0, d[1]), (c[0], 0.0, c[1]), (b[0], 0.0, b[1]), (a[0], 0.0, a[1])])
    return quads


def _assemble(quads, names, materials, emitters, cam):
    pos, idx, tmat, tem = [], [], [], []
    for verts, mat, em in quads:
        base = len(pos)
        pos.extend(verts)
        idx.extend([(base, base + 1, base + 2), (base, base + 2, base + 3)])  # obj.cpp: "f 1 2 3 4" → fan
        tmat.extend([names.index(mat)] * 2)
        tem.extend([em] * 2)
    return SceneDesc(np.array(pos, np.float32), np.array(idx, np.uint32), np.array(tmat, np.uint32),
                     np.array(tem, np.int32), materials, emitters, cam)


def cbox_scene(width=512, height=512):
    f32 = np.float32
    quads = []  # (four vertices, material name, emitter index)
    # luminaire: <rotate x="1" angle="180"/> then <translate y="1020" z="550"/>  →  (x, 1020 - y, 550 - z)
    lum = [(343.0, 548.79999, 227.0), (343.0, 548.79999, 332.0), (213.0, 548.79999, 332.0), (213.0, 548.79999, 227.0)]
    lum = [(x, float(f32(1020.0) - f32(y)), float(f32(550.0) - f32(z))) for (x, y, z) in lum]
    quads.append((lum, "light", 0))
    quads.append(([(552.79999, 0, 0), (0, 0, 0), (0, 0, 559.20001), (549.59998, 0, 559.20001)], "white", -1))  # floor
    quads.append(([(556, 548.79999, 0), (556, 548.79999, 559.20001), (0, 548.79999, 559.20001), (0, 548.79999, 0)], "white", -1))  # ceiling
    quads.append(([(549.59998, 0, 559.20001), (0, 0, 559.20001), (0, 548.79999, 559.20001), (556, 548.79999, 559.20001)], "white", -1))  # back
    quads.append(([(0, 0, 559.20001), (0, 0, 0), (0, 548.79999, 0), (0, 548.79999, 559.20001)], "green", -1))
    quads.append(([(552.79999, 0, 0), (549.59998, 0, 559.20001), (556, 548.79999, 559.20001), (556, 548.79999, 0)], "red", -1))
    for q in _extruded_box([(130.0, 65.0), (82.0, 225.0), (240.0, 272.0), (290.0, 114.0)], 165.0):
        quads.append((q, "box", -1))
    for q in _extruded_box([(423.0, 247.0), (265.0, 296.0), (314.0, 456.0), (472.0, 406.0)], 330.0):
        quads.append((q, "box", -1))
    names = ["box", "white", "red", "green", "light"]
    materials = [dict(type=0, reflectance=CBOX_RGB[n]) for n in names]
    emitters = [dict(radiance=CBOX_EMITTER_RGB)]
    cam = perspective_camera((278, 273, -800), (278, 273, -799), (0, 1, 0), 39.3077, "smaller", 10.0, 2800.0, width, height)
    return _assemble(quads, names, materials, emitters, cam)


def room_scene(width=1280, height=720, n_boxes=2000, tess=4, seed=1234, glossy=False):
    """Kitchen-class stand-in: 4 x 3 x 5 m closed room, one emitter behind a
    ceiling slit (all light is indirect), `n_boxes` random boxes each face tessellated tess x tess.
    Triangles = 12 * tess^2 * n_boxes + 24.  Deterministic in `seed`.  Lambertian only, or with glossy=True the S3 material mix:
    one box material in three becomes GGX (alpha 0.1) metal, one plastic, the floor plastic (same geometry)."""
    rng = np.random.RandomState(seed)
    mats = ["floor", "wall", "left", "right", "light", "b0", "b1", "b2"]
    quads, qmat, qem = [], [], []  # arrays of shape [n, 4, 3]

    def add_quads(q, mat, em=-1):
        q = np.asarray(q, np.float64).reshape(-1, 4, 3)
        quads.append(q)
        qmat.append(np.full(len(q), mats.index(mat), np.uint32))
        qem.append(np.full(len(q), em, np.int32))

    X, Y, Z = 4.0, 3.0, 5.0
    add_quads([(0, 0, 0), (0, 0, Z), (X, 0, Z), (X, 0, 0)], "floor")
    add_quads([(0, 0, Z), (0, Y, Z), (X, Y, Z), (X, 0, Z)], "wall")       # back (z = Z), faces -z
    add_quads([(0, 0, 0), (X, 0, 0), (X, Y, 0), (0, Y, 0)], "wall")       # front (z = 0), faces +z
    add_quads([(0, 0, 0), (0, Y, 0), (0, Y, Z), (0, 0, Z)], "left")       # x = 0, faces +x
    add_quads([(X, 0, 0), (X, 0, Z), (X, Y, Z), (X, Y, 0)], "right")      # x = X, faces -x
    # ceiling with a slit along x at z in [2.3, 2.7]; light box above the slit
    add_quads([(0, Y, 0), (X, Y, 0), (X, Y, 2.3), (0, Y, 2.3)], "wall")
    add_quads([(0, Y, 2.7), (X, Y, 2.7), (X, Y, Z), (0, Y, Z)], "wall")
    add_quads([(0, Y + 0.5, 2.3), (X, Y + 0.5, 2.3), (X, Y + 0.5, 2.7), (0, Y + 0.5, 2.7)], "light", 0)  # emitter, faces down
    add_quads([(0, Y, 2.3), (X, Y, 2.3), (X, Y + 0.5, 2.3), (0, Y + 0.5, 2.3)], "wall")
    add_quads([(0, Y, 2.7), (0, Y + 0.5, 2.7), (X, Y + 0.5, 2.7), (X, Y, 2.7)], "wall")
    add_quads([(0, Y, 2.3), (0, Y + 0.5, 2.3), (0, Y + 0.5, 2.7), (0, Y, 2.7)], "wall")
    add_quads([(X, Y, 2.3), (X, Y, 2.7), (X, Y + 0.5, 2.7), (X, Y + 0.5, 2.3)], "wall")

    ii, jj = np.meshgrid(np.arange(tess), np.arange(tess), indexing="ij")
    u0, u1, v0, v1 = (ii / tess).ravel(), ((ii + 1) / tess).ravel(), (jj / tess).ravel(), ((jj + 1) / tess).ravel()
    for k in range(n_boxes):
        sx, sz = rng.uniform(0.05, 0.25, 2)
        sy = rng.uniform(0.05, 0.9)
        cx, cz = rng.uniform(0.3, X - 0.3), rng.uniform(1.2, Z - 0.3)
        y0 = 0.0 if rng.rand() < 0.7 else rng.uniform(0.3, 2.0)
        x0, x1, z0, z1, y1 = cx - sx, cx + sx, cz - sz, cz + sz, y0 + sy
        faces = np.array([
            [(x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0)],  # top (+y)
            [(x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)],  # bottom (-y)
            [(x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0)],  # -z
            [(x1, y0, z1), (x1, y1, z1), (x0, y1, z1), (x0, y0, z1)],  # +z
            [(x0, y0, z1), (x0, y1, z1), (x0, y1, z0), (x0, y0, z0)],  # -x
            [(x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)],  # +x
        ], np.float64)
        a, b, d = faces[:, 0][:, None, :], faces[:, 1][:, None, :], faces[:, 3][:, None, :]
        P = lambda u, v: a + (b - a) * u[None, :, None] + (d - a) * v[None, :, None]  # noqa: E731
        sub = np.stack([P(u0, v0), P(u1, v0), P(u1, v1), P(u0, v1)], axis=2)  # [6, tess^2, 4, 3]
        add_quads(sub.reshape(-1, 4, 3), "b%d" % (k % 3))
    q = np.concatenate(quads).astype(np.float32)
    n = len(q)
    positions = q.reshape(-1, 3)
    base = (np.arange(n, dtype=np.uint32) * 4)[:, None]
    indices = np.concatenate([base + np.array([0, 1, 2], np.uint32), base + np.array([0, 2, 3], np.uint32)], axis=1).reshape(-1, 3)
    tri_mat = np.repeat(np.concatenate(qmat), 2)
    tri_em = np.repeat(np.concatenate(qem), 2)
    refl = {"floor": (0.6, 0.55, 0.5), "wall": (0.75, 0.75, 0.75), "left": (0.6, 0.1, 0.1), "right": (0.1, 0.5, 0.15),
            "light": (0.0, 0.0, 0.0), "b0": (0.7, 0.6, 0.4), "b1": (0.3, 0.4, 0.7), "b2": (0.8, 0.8, 0.8)}
    materials = [dict(type=0, reflectance=refl[m]) for m in mats]
    if glossy:
        materials[mats.index("b0")] = dict(type="roughconductor", alpha=0.1, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), reflectance=(1.0, 1.0, 1.0))
        materials[mats.index("b1")] = dict(type="plastic", reflectance=refl["b1"], specular=(1.0, 1.0, 1.0), eta=1.49)
        materials[mats.index("floor")] = dict(type="plastic", reflectance=refl["floor"], specular=(1.0, 1.0, 1.0), eta=1.49)
    emitters = [dict(radiance=(60.0, 55.0, 45.0))]
    cam = perspective_camera((2.0, 1.5, 0.15), (2.0, 1.2, 3.0), (0, 1, 0), 70.0, "x", 0.05, 100.0, width, height)
    return SceneDesc(positions, indices, tri_mat.astype(np.uint32), tri_em.astype(np.int32), materials, emitters, cam)


def torus_scene(width=1920, height=1080, n_major=96, n_minor=48):
    """Torus-class stand-in (the paper's TORUS scene is not bundled with the reference): a diffuse torus inside a
    glass cube (dielectric, eta 1.5) on a diffuse floor in a closed room, lit by one small area emitter — every path that reaches the
    torus is specular-diffuse-specular.  2 * n_major * n_minor + 12 (cube) + 12 (room) + 2 (lamp) triangles, deterministic."""
    P, I, M, E = [], [], [], []

    def quad(a, b, c, d, mat, em=-1):
        k = len(P)
        P.extend([a, b, c, d]); I.extend([(k, k + 1, k + 2), (k, k + 2, k + 3)]); M.extend([mat, mat]); E.extend([em, em])

    def box(lo, hi, mat, inward):
        x0, y0, z0 = lo; x1, y1, z1 = hi
        faces = [((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)),   # bottom (normal -y when outward)
                 ((x0, y1, z0), (x0, y1, z1), (x1, y1, z1), (x1, y1, z0)),   # top
                 ((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),   # -x
                 ((x1, y0, z0), (x1, y1, z0), (x1, y1, z1), (x1, y0, z1)),   # +x
                 ((x0, y0, z0), (x0, y1, z0), (x1, y1, z0), (x1, y0, z0)),   # -z
                 ((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1))]   # +z
        for f in faces:
            quad(*(f[::-1] if inward else f), mat)
    box((-4.0, 0.0, -4.0), (4.0, 4.0, 4.0), 0, True)            # the room, normals inward
    box((-1.0, 0.002, -1.0), (1.0, 1.2, 1.0), 1, False)         # the glass cube (a hair above the floor)
    R, r, cy = 0.55, 0.2, 0.45                                  # the torus, axis +y
    base = len(P)
    for a in range(n_major):
        ua = 2 * math.pi * a / n_major
        for b in range(n_minor):
            vb = 2 * math.pi * b / n_minor
            rr = R + r * math.cos(vb)
            P.append((rr * math.cos(ua), cy + r * math.sin(vb), rr * math.sin(ua)))
    for a in range(n_major):
        for b in range(n_minor):
            i00 = base + a * n_minor + b; i01 = base + a * n_minor + (b + 1) % n_minor
            i10 = base + ((a + 1) % n_major) * n_minor + b; i11 = base + ((a + 1) % n_major) * n_minor + (b + 1) % n_minor
            I.extend([(i00, i01, i11), (i00, i11, i10)]); M.extend([2, 2]); E.extend([-1, -1])
    quad((-0.15, 3.6, -0.15), (0.15, 3.6, -0.15), (0.15, 3.6, 0.15), (-0.15, 3.6, 0.15), 3, 0)   # the lamp, facing down
    materials = [dict(type=0, reflectance=(0.6, 0.6, 0.6)), dict(type="dielectric", eta=1.5, reflectance=(1, 1, 1), specular=(1, 1, 1)),
                 dict(type=0, reflectance=(0.8, 0.35, 0.15)), dict(type=0, reflectance=(0, 0, 0))]
    cam = perspective_camera((3.3, 2.5, -3.6), (0.0, 0.5, 0.0), (0, 1, 0), 46.0, "x", 0.01, 100.0, width, height)
    return SceneDesc(np.array(P, np.float32), np.array(I, np.uint32), np.array(M, np.uint32), np.array(E, np.int32), materials,
                     [dict(radiance=(400.0, 400.0, 400.0))], cam)
