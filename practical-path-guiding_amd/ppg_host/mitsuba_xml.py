"""Loader for the subset of Mitsuba 0.6 scene XML the reference's bundled scenes use:

  integrator  guided_path (every property of GP:1014-1085 and integrator.cpp:192-218)
  sensor      perspective (fov, fovAxis, nearClip, farClip, toWorld; focusDistance ignored: pinhole),
              nested sampler (independent; sampleCount / seed do not steer guided_path, GP:1342-1374) and
              film hdrfilm (width, height; rfilter box)
  shapes      obj (filename, toWorld, faceNormals, maxSmoothAngle, flipNormals, flipTexCoords, collapse),
              serialized (filename, shapeIndex, toWorld, faceNormals, maxSmoothAngle, flipNormals), ply (filename, toWorld, faceNormals, maxSmoothAngle, flipNormals), rectangle / cube (toWorld, flipNormals),
              sphere (center, radius, toWorld = rotation x uniform scale, flipNormals) — analytic, not tessellated
  bsdfs       diffuse, conductor (material none, explicit eta / k, or a named material read from Mitsuba's data/ior), roughconductor / roughdielectric / roughplastic (ggx / beckmann, isotropic; roughplastic reads Mitsuba's data/microfacet tables),
              plastic, dielectric, thindielectric,
              mask (constant opacity), twosided(any of the BRDFs) — top level with id, nested, or <ref id>
  emitters    area (nested in a shape), constant (environment), envmap (latitude-longitude .exr / .pfm / .hdr; filename, scale, toWorld = rotation)
  values      <spectrum>, <rgb>, <srgb>, <blackbody> (spectrum.py), <transform> of translate / rotate / scale / lookAt / matrix,
              <default name value> and $name substitution (mitsuba -D, mitsuba.cpp:58-87)

Anything else raises SceneError naming the plugin (Mitsuba would load it; this path cannot render it yet —
SURVEY.md §8(f1)/(f2)).  With strict=False unsupported BSDFs become diffuse(0.5) and are listed in `warnings`.

The OBJ reader follows shapes/obj.cpp:198-342 (fan triangulation of n-gons, negative indices, one mesh per `g` /
`usemtl` run, vertices merged per mesh on identical (position, normal, uv), normals transformed by the inverse
transpose) and TriMesh::computeNormals (trimesh.cpp:608-676: angle-weighted vertex normals when the file has none
and faceNormals is off).  Arithmetic is float32 like the reference's SINGLE_PRECISION build.
"""
import math
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

from . import spectrum
from .scenes import SceneDesc, perspective_camera_from_matrix

f32 = np.float32
GUIDED_PATH_PROPS = {  # name → type (GP:1014-1085, integrator.cpp:192-218)
    "nee": str, "sampleCombination": str, "spatialFilter": str, "directionalFilter": str, "bsdfSamplingFractionLoss": str,
    "budgetType": str, "sdTreeMaxMemory": int, "sTreeThreshold": int, "dTreeThreshold": float, "bsdfSamplingFraction": float,
    "sppPerPass": int, "budget": float, "dumpSDTree": int, "rrDepth": int, "maxDepth": int, "strictNormals": int, "hideEmitters": int,
}


class SceneError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------- transforms
def _translate(x, y, z):
    m = np.eye(4, dtype=f32)
    m[:3, 3] = (x, y, z)
    return m


def _scale(x, y, z):
    return np.diag(np.array([x, y, z, 1], f32))


def _rotate(axis, angle_deg):  # Transform::rotate, transform.cpp:65-97
    a = np.asarray(axis, f32)
    a = a / f32(np.sqrt(np.dot(a, a)))
    th = f32(angle_deg) * f32(math.pi / 180.0)
    s, c = f32(np.sin(th)), f32(np.cos(th))
    x, y, z = a
    one = f32(1)
    m = np.eye(4, dtype=f32)
    m[0, :3] = (x * x + (one - x * x) * c, x * y * (one - c) - z * s, x * z * (one - c) + y * s)
    m[1, :3] = (x * y * (one - c) + z * s, y * y + (one - y * y) * c, y * z * (one - c) - x * s)
    m[2, :3] = (x * z * (one - c) - y * s, y * z * (one - c) + x * s, z * z + (one - z * z) * c)
    return m


def _look_at(origin, target, up):  # Transform::lookAt, transform.cpp:191-214
    p, t, u = (np.asarray(v, f32) for v in (origin, target, up))
    d = t - p
    d = d / f32(np.sqrt(np.dot(d, d)))
    left = np.cross(u, d).astype(f32)
    left = left / f32(np.sqrt(np.dot(left, left)))
    new_up = np.cross(d, left).astype(f32)
    m = np.eye(4, dtype=f32)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, p
    return m


def _floats(text):
    return [float(v) for v in text.replace(",", " ").split()]


def _transform(elem, sub):
    m = np.eye(4, dtype=f32)
    for c in elem:
        g = lambda k, d: float(sub(c.get(k, str(d))))  # noqa: E731
        if c.tag == "translate":
            t = _translate(g("x", 0), g("y", 0), g("z", 0))
        elif c.tag == "scale":
            t = _scale(*([g("value", 1)] * 3)) if c.get("value") is not None else _scale(g("x", 1), g("y", 1), g("z", 1))
        elif c.tag == "rotate":
            t = _rotate((g("x", 0), g("y", 0), g("z", 0)), g("angle", 0))
        elif c.tag == "lookAt" or c.tag == "lookat":
            o, tg = _floats(sub(c.get("origin"))), _floats(sub(c.get("target")))
            up = _floats(sub(c.get("up"))) if c.get("up") else None
            if up is None:  # scenehandler.cpp: any vector orthogonal to the viewing direction
                d = np.asarray(tg) - np.asarray(o)
                up = np.cross(d, (1, 0, 0) if abs(d[0]) < abs(d[1]) else (0, 1, 0))
            t = _look_at(o, tg, up)
        elif c.tag == "matrix":
            v = _floats(sub(c.get("value")))
            if len(v) != 16:
                raise SceneError("<matrix> needs 16 values")
            t = np.asarray(v, f32).reshape(4, 4)
        else:
            raise SceneError("unsupported transform element <%s>" % c.tag)
        m = (t @ m).astype(f32)  # scenehandler.cpp: later elements are applied after earlier ones
    return m


def _xf_points(m, p):
    q = p @ m[:3, :3].T + m[:3, 3]
    w = p @ m[3, :3] + m[3, 3]
    return np.where((w != 1)[:, None], q / w[:, None], q).astype(f32)


def _xf_normals(m, n):
    inv_t = np.linalg.inv(m.astype(np.float64))[:3, :3].T.astype(f32)  # Transform::operator()(Normal): inverse transpose
    return (n @ inv_t.T).astype(f32)


# ---------------------------------------------------------------------------------------------- OBJ
def _fetch_lines(path):
    with open(path, "r", errors="replace") as f:
        pending = ""
        for raw in f:
            line = pending + raw.rstrip(" \t\r\n")
            if line.endswith("\\"):  # obj.cpp:166-188: a trailing backslash continues the line
                pending = line[:-1]
                continue
            pending = ""
            yield line
        if pending:
            yield pending


def _unit_angle(u, v):  # util.h:309-314
    d = np.sum(u * v, 1)
    return np.where(d < 0, f32(math.pi) - f32(2) * np.arcsin(f32(0.5) * np.linalg.norm(v + u, axis=1).astype(f32)),
                    f32(2) * np.arcsin(f32(0.5) * np.linalg.norm(v - u, axis=1).astype(f32))).astype(f32)


def compute_normals(pos, tris, flip=False):
    """TriMesh::computeNormals, smooth branch (trimesh.cpp:631-671): angle-weighted face normals, triangle-major."""
    n = np.zeros_like(pos, dtype=f32)
    p = pos[tris]  # (T, 3, 3)
    a, b = p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]
    fn = np.cross(a, b).astype(f32)
    ln = np.sqrt(np.sum(fn * fn, 1)).astype(f32)
    ok = ln != 0
    fn[ok] = fn[ok] / ln[ok, None]
    idx, contrib = [], []
    for i in range(3):
        v0, v1, v2 = p[:, i], p[:, (i + 1) % 3], p[:, (i + 2) % 3]
        sa, sb = (v1 - v0).astype(f32), (v2 - v0).astype(f32)
        with np.errstate(invalid="ignore", divide="ignore"):
            ua = sa / np.sqrt(np.sum(sa * sa, 1)).astype(f32)[:, None]
            ub = sb / np.sqrt(np.sum(sb * sb, 1)).astype(f32)[:, None]
            ang = _unit_angle(ua, ub)
        idx.append(tris[:, i]); contrib.append(fn * ang[:, None])
    order = np.stack(idx, 1).reshape(-1)            # triangle-major: (t0 c0, t0 c1, t0 c2, t1 c0, ...)
    vals = np.stack(contrib, 1).reshape(-1, 3)
    keep = np.repeat(ok, 3)                          # degenerate triangles contribute nothing ("break" at i == 0)
    np.add.at(n, order[keep], vals[keep])
    length = np.sqrt(np.sum(n * n, 1)).astype(f32)
    if flip:
        length = -length
    bad = length == 0
    with np.errstate(invalid="ignore", divide="ignore"):
        n = (n / length[:, None]).astype(f32)
    n[bad] = (1, 0, 0)
    return n


def _cosf(x):
    """cosf of the C library (the crease threshold of rebuild_topology must not depend on numpy's float32 cosine)."""
    import ctypes
    import ctypes.util
    global _libm_cos
    try:
        _libm_cos
    except NameError:
        _libm_cos = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _libm_cos.cosf.restype, _libm_cos.cosf.argtypes = ctypes.c_float, [ctypes.c_float]
    return f32(_libm_cos.cosf(float(f32(x))))


def rebuild_topology(pos, uvs, idx, max_angle):
    """TriMesh::rebuildTopology (trimesh.cpp:468-608): vertices are re-merged on (position, uv) and, around every such vertex, the
    incident triangles are greedily clustered by face normal — a new vertex per cluster of faces whose normals differ by less than
    `max_angle` degrees from the cluster's first face — so that the smooth normals computed afterwards stop at creases.
    Returns (positions, uvs or None, indices); new vertices are numbered in the reference's order (sorted by position, then uv;
    clusters in triangle order)."""
    pos = np.asarray(pos, f32); idx = np.asarray(idx, np.int64)
    dp_thresh = _cosf(f32(max_angle) * f32(math.pi / 180.0))  # std::cos(degToRad(maxAngle))
    v0, v1, v2 = pos[idx[:, 0]], pos[idx[:, 1]], pos[idx[:, 2]]
    n = np.cross(v1 - v0, v2 - v0).astype(f32)
    ln = np.sqrt(np.sum(n * n, 1, dtype=f32), dtype=f32)
    ok = ln > f32(2.93873587705571876e-39)  # RCPOVERFLOW_FLT
    fn = np.zeros_like(n)
    fn[ok] = n[ok] / ln[ok, None]
    # the multimap<Vertex, TopoData>: key order (p.x, p.y, p.z[, uv.x, uv.y]), equal keys in insertion order (triangle, corner)
    groups = {}
    for t in range(len(idx)):
        for j in range(3):
            v = int(idx[t, j])
            key = tuple(float(c) for c in pos[v]) + (tuple(float(c) for c in uvs[v]) if uvs is not None else ())
            groups.setdefault(key, []).append(t)
    new_pos, new_uv = [], []
    new_idx = np.full(idx.shape, -1, np.int64)
    for key in sorted(groups):
        entries = groups[key]
        p = np.asarray(key[:3], f32)
        clustered = [False] * len(entries)
        for a, t1 in enumerate(entries):
            if clustered[a]:
                continue
            n1 = fn[t1]
            vertex = len(new_pos)
            new_pos.append(key[:3])
            if uvs is not None:
                new_uv.append(key[3:])
            for b in range(a, len(entries)):
                if clustered[b]:
                    continue
                n2 = fn[entries[b]]
                if np.array_equal(n1, n2) or f32(f32(n1[0] * n2[0]) + f32(n1[1] * n2[1])) + f32(n1[2] * n2[2]) > dp_thresh:
                    for i in range(3):
                        if np.array_equal(pos[idx[entries[b], i]], p):
                            new_idx[entries[b], i] = vertex
                    clustered[b] = True
    assert (new_idx >= 0).all()
    return (np.asarray(new_pos, f32).reshape(-1, 3), np.asarray(new_uv, f32).reshape(-1, 2) if uvs is not None else None,
            new_idx.astype(np.uint32))


def load_obj(path, to_world=None, face_normals=False, flip_normals=False, flip_tex_coords=True, collapse=False, max_smooth_angle=None):
    """→ list of meshes dict(name, material, positions (V,3), normals (V,3) or None, indices (T,3)) in world space."""
    to_world = np.eye(4, dtype=f32) if to_world is None else to_world
    V, N, UV = [], [], []
    tris, meshes = [], []
    material = ""
    name = os.path.splitext(os.path.basename(path))[0]

    def corner(tok):
        parts = tok.split("/")
        p = int(parts[0])
        uv = int(parts[1]) if len(parts) >= 2 and parts[1] else 0
        n = int(parts[2]) if len(parts) == 3 and parts[2] else 0
        if len(parts) > 3:
            raise SceneError("%s: invalid OBJ face format %r" % (path, tok))
        return p, uv, n

    def flush(mesh_name):
        nonlocal tris
        if not tris:
            return
        pos_w = _xf_points(to_world, np.asarray(V, f32).reshape(-1, 3)) if V else np.zeros((0, 3), f32)
        nrm_w = None
        if N:
            nrm_w = _xf_normals(to_world, np.asarray(N, f32).reshape(-1, 3))
            ln = np.sqrt(np.sum(nrm_w * nrm_w, 1)).astype(f32)
            nz = ln != 0
            nrm_w[nz] = nrm_w[nz] / ln[nz, None]
        uvs = np.asarray(UV, f32).reshape(-1, 2) if UV else np.zeros((0, 2), f32)
        vmap, vp, vn, vuv, idx = {}, [], [], [], []
        has_normals = has_uvs = False
        for t in tris:
            tri = []
            for (p, uv, n) in t:
                if p < 0: p += len(V) + 1
                if n < 0: n += len(N) + 1
                if uv < 0: uv += len(UV) + 1
                if p <= 0 or p > len(V):
                    raise SceneError("%s: vertex index %d out of bounds (max %d)" % (path, p, len(V)))
                if n > len(N) or uv > len(UV):
                    raise SceneError("%s: normal / uv index out of bounds" % path)
                pn = tuple(nrm_w[n - 1]) if n else (0.0, 0.0, 0.0)
                has_normals |= bool(n)
                has_uvs |= bool(uv)
                puv = tuple(uvs[uv - 1]) if uv else (0.0, 0.0)
                key = (tuple(pos_w[p - 1]), pn, puv)
                k = vmap.get(key)
                if k is None:
                    k = vmap[key] = len(vp)
                    vp.append(key[0]); vn.append(pn); vuv.append(puv)
                tri.append(k)
            idx.append(tri)
        pos = np.asarray(vp, f32).reshape(-1, 3)
        idx = np.asarray(idx, np.uint32).reshape(-1, 3)
        normals = np.asarray(vn, f32).reshape(-1, 3) if has_normals else None
        muv = np.asarray(vuv, f32).reshape(-1, 2) if has_uvs else None
        if max_smooth_angle is not None:  # obj.cpp:336-343: the file's normals are discarded, creases found from the dihedral angles
            pos, muv, idx = rebuild_topology(pos, muv, idx, max_smooth_angle)
            normals = None
        normals, idx = _finish_normals(pos, normals, idx, face_normals, flip_normals)
        meshes.append(dict(name=mesh_name, material=material, positions=pos, normals=normals, indices=idx, uvs=muv))
        tris = []

    for line in _fetch_lines(path):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            V.append([float(v) for v in tok[1:4]])
        elif tok[0] == "vn":
            N.append([float(v) for v in tok[1:4]])
        elif tok[0] == "vt":
            u, v = float(tok[1]), float(tok[2]) if len(tok) > 2 else 0.0
            UV.append([u, 1 - v if flip_tex_coords else v])
        elif tok[0] == "g" and not collapse:
            flush(name)
            name = line[1:].strip()
        elif tok[0] == "usemtl":
            if not collapse:
                flush(name)
            material = line[6:].strip()
        elif tok[0] == "f":
            c = [corner(t) for t in tok[1:]]
            if len(c) < 3:
                raise SceneError("%s: face with fewer than 3 vertices" % path)
            for k in range(1, len(c) - 1):  # n-gons as a fan (obj.cpp:322-334)
                tris.append((c[0], c[k], c[k + 1]))
    flush(name)
    return meshes


def rectangle_mesh(to_world, flip_normals=False):
    """shapes/rectangle.cpp:170-203 (createTriMesh)."""
    m = to_world
    if flip_normals:
        m = (m @ _scale(1, 1, -1)).astype(f32)  # rectangle.cpp:82-83
    pos = _xf_points(m, np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], f32))
    n = _xf_normals(m, np.array([[0, 0, 1]], f32))[0]
    n = (n / f32(np.sqrt(np.dot(n, n)))).astype(f32)
    return dict(name="rectangle", material="", positions=pos, normals=np.repeat(n[None], 4, 0), indices=np.array([[0, 1, 2], [2, 3, 0]], np.uint32))


# ---------------------------------------------------------------------------------------------- scene
def _finish_normals(pos, normals, idx, face_normals, flip_normals):
    """TriMesh::computeNormals (trimesh.cpp:608-676) as configure() applies it to a freshly loaded mesh."""
    if face_normals:
        normals = None
        if flip_normals:
            idx = idx[:, [1, 0, 2]]
    elif normals is not None:
        if flip_normals:
            normals = -normals
    else:
        normals = compute_normals(pos, idx, flip_normals)
    return normals, idx


def load_serialized(path, to_world=None, shape_index=0, face_normals=False, flip_normals=False, max_smooth_angle=None):
    """shapes/serialized.cpp:148-212 + TriMesh::loadCompressed (trimesh.cpp:175-295): Mitsuba's binary mesh format — header 0x041C,
    version 3 or 4, then a zlib stream {flags, [name], vertex count, triangle count, positions, [normals], [texcoords], [colours],
    indices}; several meshes per file are addressed through the offset table at the end (`shapeIndex`)."""
    import struct
    import zlib
    to_world = np.eye(4, dtype=f32) if to_world is None else to_world
    buf = open(path, "rb").read()

    def header(off):
        fmt, version = struct.unpack_from("<HH", buf, off)
        if fmt != 0x041C:
            raise SceneError("%s: encountered an invalid file format!" % path)
        if version not in (3, 4):
            raise SceneError("%s: encountered an incompatible file version!" % path)
        return version
    version = header(0)
    start = 0
    if shape_index != 0:  # readOffset, trimesh.cpp:272-295
        if len(buf) < 8:
            raise SceneError("%s: truncated file" % path)
        count = struct.unpack_from("<I", buf, len(buf) - 4)[0]
        if shape_index < 0 or shape_index >= count:  # (the reference accepts shapeIndex == count and then reads past the offset table)
            raise SceneError("%s: shape index is out of range! (requested %d out of 0..%d)" % (path, shape_index, count - 1))
        back = (8 if version == 4 else 4) * (count - shape_index) + 4
        if back > len(buf):
            raise SceneError("%s: corrupt offset table" % path)
        start = struct.unpack_from("<Q" if version == 4 else "<I", buf, len(buf) - back)[0]
        if start + 4 > len(buf):
            raise SceneError("%s: corrupt offset table" % path)
        header(start)
    try:
        data = zlib.decompressobj().decompress(buf[start + 4:])
    except zlib.error as e:
        raise SceneError("%s: %s" % (path, e))
    off = 0
    flags = struct.unpack_from("<I", data, off)[0]; off += 4
    if version == 4:
        off = data.index(b"\0", off) + 1
    if off + 16 > len(data):
        raise SceneError("%s: truncated mesh data" % path)
    nv, nt = struct.unpack_from("<QQ", data, off); off += 16
    if nv > len(data) or nt > len(data):
        raise SceneError("%s: truncated mesh data" % path)
    dt = "<f8" if flags & 0x2000 else "<f4"

    def take(n_comp):
        nonlocal off
        if off + nv * n_comp * (8 if flags & 0x2000 else 4) > len(data):
            raise SceneError("%s: truncated mesh data" % path)
        a = np.frombuffer(data, dt, nv * n_comp, off).reshape(nv, n_comp).astype(f32)
        off += a.shape[0] * n_comp * (8 if flags & 0x2000 else 4)
        return a
    pos = take(3)
    normals = take(3) if flags & 0x0001 else None
    uvs = take(2) if flags & 0x0002 else None
    if flags & 0x0008:
        take(3)  # vertex colours: not used
    if off + nt * 12 > len(data):
        raise SceneError("%s: truncated mesh data" % path)
    idx = np.frombuffer(data, "<u4", nt * 3, off).reshape(nt, 3).astype(np.uint32)
    if idx.size and idx.max() >= nv:
        raise SceneError("%s: vertex index out of bounds" % path)
    if not np.array_equal(to_world, np.eye(4, dtype=f32)):
        pos = _xf_points(to_world, pos)
        if normals is not None:
            normals = _xf_normals(to_world, normals)
            normals = (normals / np.sqrt(np.sum(normals * normals, 1, dtype=f32), dtype=f32)[:, None]).astype(f32)
    if np.linalg.det(to_world[:3, :3].astype(np.float64)) < 0:
        idx = idx[:, [1, 0, 2]]
    if max_smooth_angle is not None:
        pos, uvs, idx = rebuild_topology(pos, uvs, idx, max_smooth_angle)
        normals = None
    normals, idx = _finish_normals(pos, normals, idx, face_normals, flip_normals)
    return dict(name=os.path.splitext(os.path.basename(path))[0], material="", positions=pos, normals=normals, indices=idx)


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4",
              "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def load_ply(path, to_world=None, face_normals=False, flip_normals=False, max_smooth_angle=None):
    """shapes/ply.cpp: Stanford PLY (ascii / binary little / big endian) with vertex x y z [nx ny nz] [u v | s t] and faces of 3 or 4 indices
    (a quad becomes (0, 1, 2), (3, 0, 2), ply.cpp:299-311); positions and normals go through toWorld as they are read (:229-233)."""
    to_world = np.eye(4, dtype=f32) if to_world is None else to_world
    buf = open(path, "rb").read()
    end = buf.find(b"end_header")
    if not buf.startswith(b"ply") or end < 0:
        raise SceneError("%s: not a PLY file" % path)
    head = buf[:end].decode("latin1").splitlines()
    data_off = buf.index(b"\n", end) + 1
    fmt, elements = None, []
    for line in head[1:]:
        tok = line.split()
        if not tok or tok[0] in ("comment", "obj_info"):
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            elements.append([tok[1], int(tok[2]), []])
        elif tok[0] == "property":
            elements[-1][2].append((tok[-1], tok[1:-1]))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise SceneError("%s: unknown PLY format %r" % (path, fmt))
    verts, faces = {}, []
    if fmt == "ascii":
        tokens = iter(buf[data_off:].split())
        nxt = lambda typ: (float if _PLY_TYPES[typ][0] == "f" else int)(next(tokens))  # noqa: E731
    else:
        bo = "<" if fmt == "binary_little_endian" else ">"
        pos = [data_off]

        def nxt(typ):
            dt = np.dtype(bo + _PLY_TYPES[typ])
            v = np.frombuffer(buf, dt, 1, pos[0])[0]
            pos[0] += dt.itemsize
            return v
    for name, count, props in elements:
        cols = {pn: [] for pn, _ in props} if name == "vertex" else None
        for _ in range(count):
            for pn, ptype in props:
                if ptype[0] == "list":
                    n = int(nxt(ptype[1]))
                    vals = [int(nxt(ptype[2])) for _ in range(n)]
                    if name == "face" and pn in ("vertex_indices", "vertex_index"):
                        if n not in (3, 4):
                            raise SceneError("%s: only triangle and quad-based PLY meshes are supported" % path)
                        faces.append((vals[0], vals[1], vals[2]))
                        if n == 4:
                            faces.append((vals[3], vals[0], vals[2]))
                else:
                    v = nxt(ptype[0])
                    if cols is not None:
                        cols[pn].append(v)
        if name == "vertex":
            verts = cols
    if not faces or not verts or "x" not in verts:
        raise SceneError("Unable to load \"%s\" (no triangles or vertices found)!" % path)
    P = np.stack([np.asarray(verts[k], f32) for k in "xyz"], 1)
    N = np.stack([np.asarray(verts[k], f32) for k in ("nx", "ny", "nz")], 1) if all(k in verts for k in ("nx", "ny", "nz")) else None
    uvs = None
    for a, b in (("u", "v"), ("s", "t")):
        if a in verts and b in verts:
            uvs = np.stack([np.asarray(verts[a], f32), np.asarray(verts[b], f32)], 1)
    idx = np.asarray(faces, np.uint32)
    if idx.max() >= len(P):
        raise SceneError("%s: vertex index out of bounds" % path)
    pos_w = _xf_points(to_world, P)
    nrm_w = None
    if N is not None:
        nrm_w = _xf_normals(to_world, N)
        nrm_w = (nrm_w / np.sqrt(np.sum(nrm_w * nrm_w, 1, dtype=f32), dtype=f32)[:, None]).astype(f32)
    if max_smooth_angle is not None:
        pos_w, uvs, idx = rebuild_topology(pos_w, uvs, idx, max_smooth_angle)
        nrm_w = None
    nrm_w, idx = _finish_normals(pos_w, nrm_w, idx, face_normals, flip_normals)
    return dict(name=os.path.splitext(os.path.basename(path))[0], material="", positions=pos_w, normals=nrm_w, indices=idx)


def cube_mesh(to_world, flip_normals=False):
    """shapes/cube.cpp:24-30, 73-103: the cube [-1, 1]^3 as 6 faces x 4 vertices (own normals per face) and 12 triangles.  Face order
    -y, +y, +x, +z, -x, -z; a face's corners start at the corner given below and turn counter-clockwise about the face normal;
    triangles (0, 1, 2), (3, 0, 2) per face — the layout of the reference's vertex table."""
    faces = [((0, -1, 0), (1, -1, -1)), ((0, 1, 0), (1, 1, -1)), ((1, 0, 0), (1, -1, -1)), ((0, 0, 1), (1, -1, 1)), ((-1, 0, 0), (-1, -1, 1)),
             ((0, 0, -1), (1, 1, -1))]
    P, N, I = [], [], []
    for f, (n, start) in enumerate(faces):
        n, p = np.array(n, np.float64), np.array(start, np.float64)
        for _ in range(4):
            P.append(p); N.append(n)
            p = np.cross(n, p) + np.dot(n, p) * n          # a quarter turn about the normal
        I += [(4 * f, 4 * f + 1, 4 * f + 2), (4 * f + 3, 4 * f, 4 * f + 2)]
    pos = _xf_points(to_world, np.asarray(P, f32))
    nrm = _xf_normals(to_world, np.asarray(N, f32))
    nrm = (nrm / np.sqrt(np.sum(nrm * nrm, 1, dtype=f32), dtype=f32)[:, None]).astype(f32)
    if flip_normals:  # TriMesh::computeNormals with existing normals (trimesh.cpp:612-618)
        nrm = -nrm
    return dict(name="cube", material="", positions=pos, normals=nrm, indices=np.asarray(I, np.uint32))


def _props(elem, sub):
    out = {}
    for c in elem:
        n = c.get("name")
        if c.tag == "boolean":
            out[n] = sub(c.get("value")).strip().lower() == "true"
        elif c.tag == "integer":
            out[n] = int(sub(c.get("value")))
        elif c.tag == "float":
            out[n] = float(sub(c.get("value")))
        elif c.tag == "string":
            out[n] = sub(c.get("value"))
        elif c.tag == "point":
            out[n] = tuple(float(sub(c.get(k, "0"))) for k in "xyz")
    return out


def load_scene(path, defines=None, strict=True, width=None, height=None, data_dir=None, mitsuba_src=None):
    """Parse `path` → (SceneDesc, integrator properties for ppg_create, info dict).

    defines: {"name": "value"} like `mitsuba -D name=value`; width/height override the film size; data_dir: the `data` directory of
    a Mitsuba tree (default $PPG_MITSUBA_DATA) — only `roughplastic` needs it, for data/microfacet/*.dat."""
    root = ET.parse(path).getroot()
    if root.tag != "scene":
        raise SceneError("%s: root element is <%s>, expected <scene>" % (path, root.tag))
    base = os.path.dirname(os.path.abspath(path))
    params = dict(defines or {})
    for d in root.iter("default"):
        params.setdefault(d.get("name"), d.get("value"))

    def sub(text):
        if text is None or "$" not in text:
            return text
        def rep(mo):
            k = mo.group(1)
            if k not in params:
                raise SceneError("undefined parameter $%s (pass it with -D %s=...)" % (k, k))
            return str(params[k])
        return re.sub(r"\$(\w+)", rep, text)

    warnings = []
    # ---- integrator
    integ = root.find("integrator")
    if integ is None:
        raise SceneError("no <integrator>")
    if integ.get("type") != "guided_path":
        raise SceneError("integrator type %r is not supported: this path implements 'guided_path' only" % integ.get("type"))
    props = {}
    for k, v in _props(integ, sub).items():
        if k not in GUIDED_PATH_PROPS:
            warnings.append("integrator property %r is not used by guided_path" % k)  # Mitsuba warns about unqueried properties, too
            continue
        props[k] = GUIDED_PATH_PROPS[k](v)
    # ---- sensor / film
    sensor = root.find("sensor")
    if sensor is None:
        raise SceneError("no <sensor>")
    if sensor.get("type") != "perspective":
        raise SceneError("sensor type %r is not supported (perspective only)" % sensor.get("type"))
    sp = _props(sensor, sub)
    film = sensor.find("film")
    fp = _props(film, sub) if film is not None else {}
    if film is not None and film.get("type") not in ("hdrfilm", "ldrfilm", None):
        raise SceneError("film type %r is not supported" % film.get("type"))
    rf = film.find("rfilter") if film is not None else None
    if rf is not None and rf.get("type") != "box":
        raise SceneError("rfilter type %r is not supported (box only; hdrfilm's default 'gaussian' neither)" % rf.get("type"))
    if rf is None:
        warnings.append("no <rfilter>: Mitsuba would default to gaussian; the box filter is used")
    W = int(width or fp.get("width", 768)); H = int(height or fp.get("height", 576))
    if "fov" not in sp:
        raise SceneError("perspective sensor without 'fov' (focalLength is not supported)")
    tw = sensor.find("transform")
    c2w = _transform(tw, sub) if tw is not None else np.eye(4, dtype=f32)
    camera = perspective_camera_from_matrix(c2w, sp["fov"], str(sp.get("fovAxis", "x")).lower(), sp.get("nearClip", 1e-2), sp.get("farClip", 1e4), W, H)
    sampler = sensor.find("sampler")
    info = dict(width=W, height=H, sample_count=_props(sampler, sub).get("sampleCount") if sampler is not None else None)

    # ---- bsdfs
    materials, mat_index, by_id = [], {}, {}
    tex_by_id = {t.get("id"): t for t in root.findall("texture") if t.get("id")}

    def colour(elem, name, default):
        for c in elem:
            if c.get("name") == name:
                if c.tag in ("rgb", "srgb", "spectrum"):
                    if c.get("filename"):  # InterpolatedSpectrum(path) (spectrum.cpp:575-600): "wavelength value" lines, # comments
                        fn = sub(c.get("filename"))
                        full = fn if os.path.isabs(fn) else os.path.join(base, fn)
                        if c.tag != "spectrum" or c.get("value") is not None or not os.path.exists(full):
                            raise SceneError("<%s filename=%r>: %s" % (c.tag, fn, "file not found" if c.tag == "spectrum" and c.get("value") is None
                                                                         else "please provide one of 'value' or 'filename'"))
                        pairs = spectrum.read_spd(full)
                        if len(pairs) < 2:
                            raise SceneError("<spectrum filename=%r>: fewer than two samples" % fn)
                        return spectrum.interpolated_to_rgb(pairs)
                    return spectrum.parse(c.tag, sub(c.get("value")))
                if c.tag == "blackbody":
                    t = sub(c.get("temperature", "")).strip()
                    return spectrum.blackbody_to_rgb(float(t[:-1] if t[-1:] in "kK" else t), float(sub(c.get("scale", "1"))))
                if c.tag == "texture" or c.tag == "ref":
                    if strict:
                        raise SceneError("a texture on %r is not supported (bitmaps on diffuse reflectances and bump maps are)" % name)
                    warnings.append("texture on %r ignored: the plug-in's default value is used" % name)
                    break
        return np.full(3, default, f32)

    textures, tex_index = [], {}

    def bitmap_texture(elem, what):
        """<texture type="bitmap"> (textures/bitmap.cpp:90-330) → index into `textures`.  The image is decoded here — 8-bit files through the
        sRGB curve unless `gamma` says otherwise (bitmap.cpp:182-204), luminance replicated to RGB — because the integrator only ever
        reads level 0 of the MIP map (include/ppg.h ppg_texture)."""
        if elem.tag == "ref":
            target = tex_by_id.get(elem.get("id"))
            if target is None:
                raise SceneError("<ref id=%r>: no such texture" % elem.get("id"))
            elem = target
        if elem.get("type") != "bitmap":
            raise SceneError("texture type %r on %s is not supported (bitmap only)" % (elem.get("type"), what))
        tp = _props(elem, sub)
        fn = tp.get("filename")
        if not fn:
            raise SceneError("bitmap texture without filename")
        full = fn if os.path.isabs(fn) else os.path.join(base, fn)
        wrap = str(tp.get("wrapMode", "repeat")).lower()
        wu, wv = str(tp.get("wrapModeU", wrap)).lower(), str(tp.get("wrapModeV", wrap)).lower()
        for wm in (wu, wv):
            if wm not in ("repeat", "mirror", "clamp", "zero", "one"):
                raise SceneError("bitmap: invalid wrap mode %r" % wm)  # bitmap.cpp:113-127
        filt = str(tp.get("filterType", "ewa")).lower()
        if filt not in ("ewa", "trilinear", "nearest", "bilinear"):
            raise SceneError("bitmap: invalid filter type %r" % filt)
        gamma = float(tp.get("gamma", 0.0))
        key = (full, wu, wv, filt == "nearest", gamma, float(tp.get("uscale", 1.0)), float(tp.get("vscale", 1.0)), float(tp.get("uoffset", 0.0)), float(tp.get("voffset", 0.0)),
               str(tp.get("channel", "")))
        if key in tex_index:
            return tex_index[key]
        if not os.path.exists(full):
            raise SceneError("bitmap: file '%s' not found" % full)
        if tp.get("channel"):
            raise SceneError("bitmap: the `channel` parameter is not supported")
        from . import imageio
        from .scenes import srgb8_table
        t = dict(uv_scale=(float(tp.get("uscale", 1.0)), float(tp.get("vscale", 1.0))), uv_offset=(float(tp.get("uoffset", 0.0)), float(tp.get("voffset", 0.0))),
                 wrap_u=wu, wrap_v=wv, nearest=filt == "nearest", source=os.path.basename(full))
        ext = os.path.splitext(full)[1].lower()
        if ext in (".exr", ".pfm", ".hdr", ".rgbe"):
            t["rgb"] = np.ascontiguousarray(imageio.read_image(full), f32)  # floating point data is linear (bitmap.cpp:190-192)
        else:
            try:
                from PIL import Image
            except ImportError:
                raise SceneError("bitmap: decoding '%s' needs PIL (scene conversion only)" % full)
            im = Image.open(full)
            if im.mode in ("P", "RGBA", "LA", "CMYK", "1", "I;16", "I"):  # alpha is dropped for a spectrum texture (bitmap.cpp:218-236: EAuto → RGB / luminance)
                im = im.convert("RGB" if im.mode != "LA" else "L")
            a = np.asarray(im)
            if a.dtype != np.uint8:
                raise SceneError("bitmap: '%s': only 8-bit images are decoded here" % full)
            if a.ndim == 2:
                a = np.repeat(a[:, :, None], 3, 2)
            a = np.ascontiguousarray(a[:, :, :3])
            if gamma == 0.0 or gamma == -1.0:  # the file's own encoding: sRGB for 8-bit images
                t["srgb8"] = a
                t["rgb"] = srgb8_table()[a]
            else:  # an explicit gamma (bump maps ask for 1.0, bumpmap.cpp:121-126): value^gamma on value / 255
                t["rgb"] = np.power(a.astype(np.float64) / 255.0, gamma).astype(f32)
        # Texture::getAverage: the mean of the full-resolution image (the MIP map's 1x1 level; exact for power-of-two sizes)
        t["average"] = [float(v) for v in t["rgb"].reshape(-1, 3).mean(0, dtype=np.float64)]
        tex_index[key] = len(textures)
        textures.append(t)
        return tex_index[key]

    def colour_or_texture(elem, name, default):
        """(rgb, texture index or None) of a reflectance that may be a bitmap; rgb is then the bitmap's average."""
        for c in elem:
            if c.get("name") == name and c.tag in ("texture", "ref") and (c.tag == "texture" or c.get("id") in tex_by_id):
                try:
                    ti = bitmap_texture(c, name)
                except SceneError as e:
                    if strict:
                        raise
                    warnings.append("texture on %r ignored (%s): the plug-in's default value is used" % (name, e))
                    return np.full(3, default, f32), None
                return np.asarray(textures[ti]["average"], f32), ti
        return colour(elem, name, default), None

    IOR = {"vacuum": 1.0, "air": 1.000277, "water": 1.3330, "polypropylene": 1.49, "bk7": 1.5046, "diamond": 2.419}  # well-known constants (cf. ior.h)

    def lookup_ior(p, name, default):  # lookupIOR, ior.h:95-111: a number or a material name
        v = p.get(name, default)
        if isinstance(v, (int, float)):
            return float(v)
        if str(v).lower() not in IOR:
            raise SceneError("IOR name %r is not in the built-in list %s; give a number" % (v, sorted(IOR)))
        return IOR[str(v).lower()]

    def microfacet_alpha(p, what):  # MicrofacetDistribution(props), microfacet.h:99-145
        if str(p.get("distribution", "beckmann")).lower() not in ("ggx", "beckmann"):
            raise SceneError("%s: distribution %r is not supported (ggx, beckmann)" % (what, p.get("distribution", "beckmann")))
        if "alphaU" in p or "alphaV" in p:
            if p.get("alphaU") != p.get("alphaV"):
                raise SceneError("%s: anisotropic roughness is not supported" % what)
            return float(p["alphaU"])
        if not p.get("sampleVisible", True):
            raise SceneError("%s: sampleVisible=false is not supported" % what)
        return float(p.get("alpha", 0.1))

    def conductor_ior(elem, p, what):  # conductor.cpp:160-186 / roughconductor.cpp:174-186
        ext = lookup_ior(p, "extEta", "air")
        name = str(p.get("material", "Cu"))
        if name.lower() == "none":
            int_eta, int_k = np.zeros(3, f32), np.ones(3, f32)
        else:
            have = {c.get("name") for c in elem}
            int_eta = int_k = None
            if not ({"eta", "k"} <= have):  # the measured spectra ship with Mitsuba (data/ior/<name>.{eta,k}.spd), not with this repository
                ddir = data_dir or os.environ.get("PPG_MITSUBA_DATA")
                files = [os.path.join(ddir, "ior", "%s.%s.spd" % (name, part)) for part in ("eta", "k")] if ddir else []
                if not ddir or not all(os.path.exists(f) for f in files):
                    raise SceneError("%s(material=%s): the measured IOR spectra data/ior/%s.{eta,k}.spd come with Mitsuba: pass data_dir (--data-dir) / "
                                     "PPG_MITSUBA_DATA, or give <spectrum|rgb name=\"eta\"> and \"k\"" % (what, name, name))
                int_eta, int_k = (spectrum.interpolated_to_rgb(spectrum.read_spd(f), zero_extend=False, clamp=False) for f in files)
        eta = colour(elem, "eta", 0.0) if any(c.get("name") == "eta" for c in elem) else int_eta
        k = colour(elem, "k", 1.0) if any(c.get("name") == "k" for c in elem) else int_k
        return tuple(float(v) for v in (eta / f32(ext))), tuple(float(v) for v in (k / f32(ext)))  # roughconductor.cpp:185-186

    rt_slices, rt_index = [], {}

    def make_bsdf(elem, allow_twosided=True):
        t = elem.get("type")
        p = _props(elem, sub)
        rgb = lambda name, d: tuple(float(v) for v in colour(elem, name, d))  # noqa: E731
        def textured(m, name, d):  # a bitmap on the diffuse reflectance: constant = its average, plus the texture index
            c, ti = colour_or_texture(elem, name, d)
            m["reflectance"] = tuple(float(v) for v in c)
            if ti is not None:
                m["texture"] = ti
            return m
        if t == "diffuse":
            return textured(dict(type=0), "reflectance", 0.5)
        if t == "twosided" and allow_twosided:
            inner = [c for c in elem if c.tag == "bsdf"]
            if len(inner) == 1:
                m = make_bsdf(inner[0], allow_twosided=False)
                if m["type"] in (6, 7, 8):
                    raise SceneError("twosided(dielectric): only BRDFs can be two-sided (twosided.cpp:84-88)")
                if m["type"] == 0 and not m.get("_substituted"):
                    return dict({k: v for k, v in m.items() if k in ("texture", "bump")}, type=1, reflectance=m["reflectance"])
                return dict(m, twosided=True)
            t = "twosided(%s)" % ",".join(c.get("type", "?") for c in inner)
        elif t == "mask" and allow_twosided:
            inner = [c for c in elem if c.tag == "bsdf"]
            if len(inner) == 1:
                m = make_bsdf(inner[0])
                if "opacity" in m:
                    raise SceneError("mask(mask(...)) is not supported")
                if m["type"] == 1:
                    m = dict({k: v for k, v in m.items() if k in ("texture", "bump")}, type=0, reflectance=m["reflectance"], twosided=True)
                return dict(m, opacity=rgb("opacity", 0.5))
            t = "mask(%s)" % ",".join(c.get("type", "?") for c in inner)
        elif t == "conductor":
            if str(p.get("material", "Cu")).lower() == "none":
                return dict(type=2, reflectance=rgb("specularReflectance", 1.0))
            eta, k = conductor_ior(elem, p, t)
            return dict(type=3, reflectance=rgb("specularReflectance", 1.0), eta=eta, k=k)
        elif t == "roughconductor":
            eta, k = conductor_ior(elem, p, t)
            m = dict(type=4, reflectance=rgb("specularReflectance", 1.0), eta=eta, k=k, alpha=microfacet_alpha(p, t))
            if str(p.get("distribution", "beckmann")).lower() == "beckmann":  # Mitsuba's default (microfacet.h:99)
                m["distribution"] = "beckmann"
            return m
        elif t == "plastic":
            eta = lookup_ior(p, "intIOR", "polypropylene") / lookup_ior(p, "extIOR", "air")
            return textured(dict(type=5, specular=rgb("specularReflectance", 1.0), eta=float(f32(eta)), nonlinear=bool(p.get("nonlinear", False))),
                            "diffuseReflectance", 0.5)
        elif t == "dielectric":
            eta = lookup_ior(p, "intIOR", "bk7") / lookup_ior(p, "extIOR", "air")
            return dict(type=6, reflectance=rgb("specularReflectance", 1.0), specular=rgb("specularTransmittance", 1.0), eta=float(f32(eta)))
        elif t == "thindielectric":
            eta = lookup_ior(p, "intIOR", "bk7") / lookup_ior(p, "extIOR", "air")
            return dict(type=7, reflectance=rgb("specularReflectance", 1.0), specular=rgb("specularTransmittance", 1.0), eta=float(f32(eta)))
        elif t == "roughplastic":  # roughplastic.cpp:197-227, 285-305
            from . import rtrans as _rt
            int_ior, ext_ior = lookup_ior(p, "intIOR", "polypropylene"), lookup_ior(p, "extIOR", "air")
            if int_ior < 0 or ext_ior < 0 or int_ior == ext_ior:
                raise SceneError("roughplastic: the interior and exterior indices of refraction must be positive and differ")
            alpha, eta = microfacet_alpha(p, t), float(f32(int_ior / ext_ior))
            distr = str(p.get("distribution", "beckmann")).lower()
            key = (distr, float(f32(alpha)), eta)
            if key not in rt_index:
                try:
                    rt_slices.append(_rt.roughplastic_slice(distr, alpha, eta, data_dir))
                except _rt.RoughTransmittanceError as e:
                    raise SceneError("roughplastic: %s" % e)
                rt_index[key] = len(rt_slices) - 1
            m = textured(dict(type=9, specular=rgb("specularReflectance", 1.0), eta=eta, alpha=alpha, nonlinear=bool(p.get("nonlinear", False)), rtrans=rt_index[key]),
                         "diffuseReflectance", 0.5)
            if distr == "beckmann":
                m["distribution"] = "beckmann"
            return m
        elif t == "roughdielectric":  # roughdielectric.cpp:183-211
            int_ior, ext_ior = lookup_ior(p, "intIOR", "bk7"), lookup_ior(p, "extIOR", "air")
            if int_ior < 0 or ext_ior < 0 or int_ior == ext_ior:
                raise SceneError("roughdielectric: the interior and exterior indices of refraction must be positive and differ")
            m = dict(type=8, reflectance=rgb("specularReflectance", 1.0), specular=rgb("specularTransmittance", 1.0), eta=float(f32(int_ior / ext_ior)),
                     alpha=microfacet_alpha(p, t))
            if str(p.get("distribution", "beckmann")).lower() == "beckmann":
                m["distribution"] = "beckmann"
            return m
        if t == "bumpmap":  # BumpMap (bumpmap.cpp:60-133): one nested BSDF + one displacement texture, the outermost adapter
            inner = [c for c in elem if c.tag == "bsdf"]
            disp = [c for c in elem if c.tag in ("texture", "ref") and (c.tag == "texture" or c.get("id") in tex_by_id)]
            if len(inner) == 1 and len(disp) == 1:
                m = make_bsdf(inner[0], allow_twosided)
                try:
                    if "bump" in m:
                        raise SceneError("bumpmap(bumpmap(...)) is not supported")
                    if disp[0].tag == "texture" and disp[0].get("type") == "bitmap" and "gamma" not in _props(disp[0], sub):
                        raise SceneError("When using a bitmap texture as a bump map, please explicitly specify the 'gamma' parameter of the bitmap plugin")  # bumpmap.cpp:121-126
                    return dict(m, bump=bitmap_texture(disp[0], "bumpmap"))
                except SceneError as e:
                    if strict:
                        raise
                    warnings.append("bump map dropped around its nested bsdf (%s)" % e)
                    return m
        if not strict and t in ("bumpmap", "coating", "roughcoating", "normalmap"):  # adapters around one nested BSDF: render the nested one
            inner = [c for c in elem if c.tag == "bsdf"]
            if len(inner) == 1:
                warnings.append("bsdf %r dropped around its nested bsdf" % t)
                return make_bsdf(inner[0], allow_twosided)
        if strict:
            raise SceneError("bsdf type %r is not supported yet (diffuse, conductor, roughconductor, plastic, roughplastic, dielectric, thindielectric, roughdielectric, "
                             "mask(...), twosided(...); "
                             "SURVEY.md §8 f1)" % t)
        warnings.append("bsdf %r replaced by diffuse(0.5)" % t)
        return dict(type=0, reflectance=(0.5, 0.5, 0.5), _substituted=True)

    def intern(m):
        m = {k: v for k, v in m.items() if not k.startswith("_")}
        key = tuple(sorted((k, tuple(v) if isinstance(v, (tuple, list)) else v) for k, v in m.items()))
        if key not in mat_index:
            mat_index[key] = len(materials)
            materials.append(m)
        return mat_index[key]

    for b in root.findall("bsdf"):
        if b.get("id"):
            by_id[b.get("id")] = intern(make_bsdf(b))
    # an id may also sit on a NESTED bsdf (KITCHEN references the twosided element inside a bumpmap): every element with an id is a named
    # object in Mitsuba (scenehandler.cpp: namedObjects)
    for top in root.findall("bsdf"):
        for b in top.iter("bsdf"):
            if b is not top and b.get("id") and b.get("id") not in by_id:
                by_id[b.get("id")] = intern(make_bsdf(b))
    environment = envmap = None
    for em in root.findall("emitter"):
        if em.get("type") == "constant" and environment is None and envmap is None:
            environment = tuple(float(v) for v in colour(em, "radiance", 1.0))
            continue
        if em.get("type") == "envmap" and environment is None and envmap is None:  # EnvironmentMap::EnvironmentMap, envmap.cpp:100-190
            from . import imageio
            ep = _props(em, sub)
            fn = ep.get("filename")
            if not fn:
                raise SceneError("envmap emitter without filename")
            full = fn if os.path.isabs(fn) else os.path.join(base, fn)
            if not os.path.exists(full):
                raise SceneError("envmap: file '%s' not found" % full)
            try:
                rgb = imageio.read_image(full)
            except (ValueError, AssertionError) as e:
                raise SceneError("envmap: %s" % e)
            tw = em.find("transform")
            R = (_transform(tw, sub) if tw is not None else np.eye(4, dtype=f32))[:3, :3].astype(f32)
            if not np.allclose(R @ R.T, np.eye(3), atol=1e-4):
                raise SceneError("envmap: toWorld must be a rotation")
            envmap = dict(rgb=rgb, scale=float(ep.get("scale", 1.0)), to_world=[float(v) for v in R.reshape(-1)])
            continue
        if em.get("type") == "sunsky" and environment is None and envmap is None:
            # SunSkyEmitter (sunsky.cpp:100-235) bakes sun + sky into a radiance map and instantiates `envmap` on it: same bake here
            # (ppg_host/sunsky.py); the Hosek-Wilkie / Preetham tables are read from the operator's Mitsuba source tree
            from . import sunsky
            ddir = data_dir or os.environ.get("PPG_MITSUBA_DATA")
            src = mitsuba_src or os.environ.get("PPG_MITSUBA_SRC") or (os.path.dirname(os.path.abspath(ddir)) if ddir else None)
            if not src or not os.path.exists(os.path.join(src, "src", "emitters", "sunsky", "skymodeldata.h")):
                if strict:
                    raise SceneError("sunsky: the sky model's coefficient tables (src/emitters/sunsky/skymodeldata.h, sunmodel.h) come with Mitsuba's "
                                     "source tree: pass mitsuba_src / --data-dir <tree>/data / PPG_MITSUBA_SRC")
                warnings.append("emitter 'sunsky' skipped (no Mitsuba source tree for the sky model's tables)")
                continue
            ep = _props(em, sub)
            for c in em:
                if c.get("name") == "albedo" and c.tag in ("rgb", "srgb", "spectrum"):
                    ep["albedo"] = [float(v) for v in spectrum.parse(c.tag, sub(c.get("value")))]
            try:
                rgb, _ = sunsky.bake(ep, src)
            except (ValueError, NotImplementedError) as e:
                raise SceneError("sunsky: %s" % e)
            tw = em.find("transform")
            R = (_transform(tw, sub) if tw is not None else np.eye(4, dtype=f32))[:3, :3].astype(f32)
            if not np.allclose(R @ R.T, np.eye(3), atol=1e-4):
                raise SceneError("sunsky: toWorld must be a rotation")
            envmap = dict(rgb=rgb, scale=1.0, to_world=[float(v) for v in R.reshape(-1)])
            continue
        if not strict:
            warnings.append("emitter %r skipped (not supported)" % em.get("type"))
            continue
        raise SceneError("emitter type %r is not supported (area emitters on shapes and one `constant`, `envmap` or `sunsky` environment emitter; "
                         "SURVEY.md §8 f2)" % em.get("type"))

    # ---- shapes
    collected, emitters, spheres = [], [], []
    default_mat = None
    for sh in root.findall("shape"):
        t = sh.get("type")
        sprops = _props(sh, sub)
        tw = sh.find("transform")
        m = _transform(tw, sub) if tw is not None else np.eye(4, dtype=f32)
        if t == "obj":
            fn = sprops.get("filename")
            if not fn:
                raise SceneError("obj shape without filename")
            if "shapeIndex" in sprops:
                raise SceneError("obj: shapeIndex is not supported")
            if "maxSmoothAngle" in sprops and sprops.get("faceNormals", False):
                raise SceneError("The properties 'maxSmoothAngle' and 'faceNormals' can't be specified at the same time!")  # obj.cpp:337-339
            full = fn if os.path.isabs(fn) else os.path.join(base, fn)
            if not os.path.exists(full):
                if strict:
                    raise SceneError("Wavefront OBJ file '%s' not found!" % full)  # obj.cpp:230
                warnings.append("shape skipped: Wavefront OBJ file '%s' not found" % full)
                continue
            meshes = load_obj(full, m, bool(sprops.get("faceNormals", False)), bool(sprops.get("flipNormals", False)),
                              bool(sprops.get("flipTexCoords", True)), bool(sprops.get("collapse", False)), sprops.get("maxSmoothAngle"))
        elif t == "ply":
            fn = sprops.get("filename")
            if not fn:
                raise SceneError("ply shape without filename")
            full = fn if os.path.isabs(fn) else os.path.join(base, fn)
            if not os.path.exists(full):
                if strict:
                    raise SceneError("PLY file '%s' not found" % full)
                warnings.append("shape skipped: PLY file '%s' not found" % full)
                continue
            if "maxSmoothAngle" in sprops and sprops.get("faceNormals", False):
                raise SceneError("The properties 'maxSmoothAngle' and 'faceNormals' can't be specified at the same time!")
            meshes = [load_ply(full, m, bool(sprops.get("faceNormals", False)), bool(sprops.get("flipNormals", False)), sprops.get("maxSmoothAngle"))]
        elif t == "serialized":
            fn = sprops.get("filename")
            if not fn:
                raise SceneError("serialized shape without filename")
            full = fn if os.path.isabs(fn) else os.path.join(base, fn)
            if not os.path.exists(full):
                if strict:
                    raise SceneError("serialized mesh file '%s' not found" % full)
                warnings.append("shape skipped: serialized mesh file '%s' not found" % full)
                continue
            if "maxSmoothAngle" in sprops and sprops.get("faceNormals", False):
                raise SceneError("The properties 'maxSmoothAngle' and 'faceNormals' can't be specified at the same time!")
            meshes = [load_serialized(full, m, int(sprops.get("shapeIndex", 0)), bool(sprops.get("faceNormals", False)), bool(sprops.get("flipNormals", False)),
                                      sprops.get("maxSmoothAngle"))]
        elif t == "rectangle":
            meshes = [rectangle_mesh(m, bool(sprops.get("flipNormals", False)))]
        elif t == "cube":
            meshes = [cube_mesh(m, bool(sprops.get("flipNormals", False)))]
        elif t == "sphere":  # Sphere::Sphere, sphere.cpp:108-131: the scale of toWorld goes into the radius, the rest stays a rotation
            meshes = []
            c = np.asarray(sprops.get("center", (0.0, 0.0, 0.0)), f32)
            o2w = _translate(*[float(v) for v in c])
            radius = f32(sprops.get("radius", 1.0))
            if tw is not None:
                scale = np.sqrt(np.sum(m[:3, 0] * m[:3, 0], dtype=f32), dtype=f32)  # objectToWorld(Vector(1, 0, 0)).length()
                o2w = (m @ _scale(*[float(f32(1) / scale)] * 3)).astype(f32) @ o2w
                o2w = o2w.astype(f32)
                radius = f32(radius * scale)
            if not radius > 0:
                raise SceneError("Cannot create spheres of radius <= 0")
            sphere = dict(center=tuple(float(v) for v in o2w[:3, 3]), radius=float(radius), to_world=[float(v) for v in o2w[:3, :3].reshape(-1)],
                          flip_normals=bool(sprops.get("flipNormals", False)))
        else:
            raise SceneError("shape type %r is not supported (obj, ply, serialized, rectangle, cube, sphere)" % t)
        # material: nested <bsdf> or <ref id>
        mat = None
        for c in sh:
            if c.tag == "bsdf":
                mat = intern(make_bsdf(c))
            elif c.tag == "ref":
                rid = c.get("id")
                if rid not in by_id:
                    raise SceneError("<ref id=%r>: no such bsdf" % rid)
                mat = by_id[rid]
        if mat is None:  # Shape::configure (shape.cpp:48-72): all-absorbing under an emitter, otherwise a 0.5 Lambertian "for convenience"
            if sh.find("emitter") is not None:
                mat = intern(dict(type=0, reflectance=(0.0, 0.0, 0.0)))
            else:
                if default_mat is None:
                    default_mat = intern(dict(type=0, reflectance=(0.5, 0.5, 0.5)))
                mat = default_mat
        em = -1
        e = sh.find("emitter")
        if e is not None:
            if e.get("type") != "area":
                raise SceneError("emitter type %r on a shape is not supported (area only)" % e.get("type"))
            if len(meshes) > 1:
                raise SceneError("Cannot attach an emitter to an OBJ file containing multiple objects!")  # obj.cpp:757-759
            if t == "obj" and not meshes:
                continue
            em = len(emitters)
            emitters.append(dict(radiance=tuple(float(v) for v in colour(e, "radiance", 1.0))))
        for mesh in meshes:
            collected.append((mesh, mat, em))
        if t == "sphere":
            spheres.append(dict(sphere, material=mat, emitter=em))
    if not collected and not spheres:
        raise SceneError("scene without shapes")
    # one vertex-normal array for the whole scene: a faceNormals mesh living next to smooth ones gets its vertices
    # un-shared and its face normals written out (same shading frame as "no normals": skdtree.h:388-401)
    any_normals = any(m["normals"] is not None for m, _, _ in collected)
    # texture coordinates only matter on meshes whose BSDF reads a bitmap; the others get NaN rows (= none)
    any_uvs = any(m.get("uvs") is not None and ("texture" in materials[mat] or "bump" in materials[mat]) for m, mat, _ in collected)
    uvl = []
    pos, nrm, idx, tmat, tem = [], [], [], [], []
    nv = 0
    for mesh, mat, em in collected:
        p, i, n = mesh["positions"], mesh["indices"], mesh["normals"]
        uv = mesh.get("uvs") if ("texture" in materials[mat] or "bump" in materials[mat]) else None
        if any_normals and n is None:
            if uv is not None:
                uv = uv[i.reshape(-1)]
            p = p[i.reshape(-1)]
            a, b = p[1::3] - p[0::3], p[2::3] - p[0::3]
            fnrm = np.cross(a, b).astype(f32)
            ln = np.sqrt(np.sum(fnrm * fnrm, 1)).astype(f32)
            fnrm[ln != 0] = fnrm[ln != 0] / ln[ln != 0, None]
            n = np.repeat(fnrm, 3, 0)
            i = np.arange(p.shape[0], dtype=np.uint32).reshape(-1, 3)
        T = i.shape[0]
        pos.append(p); idx.append(i + np.uint32(nv)); nrm.append(n)
        uvl.append(uv.astype(f32) if uv is not None else np.full((p.shape[0], 2), np.nan, f32))
        nv += p.shape[0]
        tmat.append(np.full(T, mat, np.uint32)); tem.append(np.full(T, em, np.int32))
    normals = np.concatenate(nrm).astype(f32) if any_normals else None
    if not collected:  # analytic spheres only
        pos, idx, tmat, tem = [np.zeros((0, 3), f32)], [np.zeros((0, 3), np.uint32)], [np.zeros(0, np.uint32)], [np.zeros(0, np.int32)]
    desc = SceneDesc(np.concatenate(pos).astype(f32), np.concatenate(idx).astype(np.uint32), np.concatenate(tmat), np.concatenate(tem),
                     materials, emitters, camera, normals, environment, np.stack(rt_slices).astype(f32) if rt_slices else None, spheres, envmap,
                     np.concatenate(uvl).astype(f32) if any_uvs else None, textures)
    info["warnings"] = warnings
    return desc, props, info


# ---------------------------------------------------------------------------------------------- writer
def save_scene_xml(desc, props, directory, name="scene"):
    """Write `desc` as <directory>/<name>.xml + meshes/*.obj in the subset above (one OBJ per material / emitter group),
    so that a procedural scene can be rendered by the reference itself — or read back by load_scene().  Returns the XML path."""
    os.makedirs(os.path.join(directory, "meshes"), exist_ok=True)
    cam = desc.camera
    c = lambda v: ", ".join(repr(float(x)) for x in v)  # noqa: E731
    out = ['<?xml version="1.0" encoding="utf-8"?>', '<scene version="0.5.0">', '\t<integrator type="guided_path">']
    for k, v in props.items():
        if k not in GUIDED_PATH_PROPS:
            continue
        t = GUIDED_PATH_PROPS[k]
        if k in ("strictNormals", "hideEmitters", "dumpSDTree"):
            out.append('\t\t<boolean name="%s" value="%s"/>' % (k, "true" if v else "false"))
        elif t is int:
            out.append('\t\t<integer name="%s" value="%d"/>' % (k, int(v)))
        elif t is float:
            out.append('\t\t<float name="%s" value="%r"/>' % (k, float(v)))
        else:
            out.append('\t\t<string name="%s" value="%s"/>' % (k, v))
    out += ['\t</integrator>', '\t<sensor type="perspective">',
            '\t\t<string name="fovAxis" value="%s"/>' % cam.get("fov_axis", "x"), '\t\t<float name="fov" value="%r"/>' % float(cam["fov"]),
            '\t\t<float name="nearClip" value="%r"/>' % float(cam["near_clip"]), '\t\t<float name="farClip" value="%r"/>' % float(cam["far_clip"]),
            '\t\t<transform name="toWorld">', '\t\t\t<matrix value="%s"/>' % " ".join(repr(float(x)) for x in np.asarray(cam["camera_to_world"]).reshape(-1)),
            '\t\t</transform>', '\t\t<sampler type="independent"/>', '\t\t<film type="hdrfilm">',
            '\t\t\t<integer name="width" value="%d"/>' % cam["width"], '\t\t\t<integer name="height" value="%d"/>' % cam["height"],
            '\t\t\t<boolean name="banner" value="false"/>', '\t\t\t<rfilter type="box"/>', '\t\t</film>', '\t</sensor>']
    from .bindings import Material
    for i, m in enumerate(desc.materials):
        t = m.get("type", 0)
        t = Material.BSDF[t] if isinstance(t, str) else int(t)
        M = Material.from_dict(m)
        R, S, E, K = (c(getattr(M, n)) for n in ("reflectance", "specular", "eta", "k"))
        one = '<float name="extIOR" value="1.0"/><float name="intIOR" value="%r"/>' % float(M.eta[0])
        body = {
            0: '<bsdf type="diffuse"%%s><rgb name="reflectance" value="%s"/></bsdf>' % R,
            1: '<bsdf type="diffuse"%%s><rgb name="reflectance" value="%s"/></bsdf>' % R,
            2: '<bsdf type="conductor"%%s><string name="material" value="none"/><rgb name="specularReflectance" value="%s"/></bsdf>' % R,
            3: '<bsdf type="conductor"%%s><float name="extEta" value="1.0"/><rgb name="eta" value="%s"/><rgb name="k" value="%s"/>'
               '<rgb name="specularReflectance" value="%s"/></bsdf>' % (E, K, R),
            4: '<bsdf type="roughconductor"%%s><string name="distribution" value="%s"/><float name="alpha" value="%r"/><float name="extEta" value="1.0"/>'
               '<rgb name="eta" value="%s"/><rgb name="k" value="%s"/><rgb name="specularReflectance" value="%s"/></bsdf>'
               % ("beckmann" if M.flags & 8 else "ggx", float(M.alpha), E, K, R),
            5: '<bsdf type="plastic"%%s>%s<rgb name="diffuseReflectance" value="%s"/><rgb name="specularReflectance" value="%s"/>'
               '<boolean name="nonlinear" value="%s"/></bsdf>' % (one, R, S, "true" if M.flags & 2 else "false"),
            6: '<bsdf type="dielectric"%%s>%s<rgb name="specularReflectance" value="%s"/><rgb name="specularTransmittance" value="%s"/></bsdf>' % (one, R, S),
            7: '<bsdf type="thindielectric"%%s>%s<rgb name="specularReflectance" value="%s"/><rgb name="specularTransmittance" value="%s"/></bsdf>' % (one, R, S),
            8: '<bsdf type="roughdielectric"%%s><string name="distribution" value="%s"/><float name="alpha" value="%r"/>%s'
               '<rgb name="specularReflectance" value="%s"/><rgb name="specularTransmittance" value="%s"/></bsdf>'
               % ("beckmann" if M.flags & 8 else "ggx", float(M.alpha), one, R, S),
            9: '<bsdf type="roughplastic"%%s><string name="distribution" value="%s"/><float name="alpha" value="%r"/>%s'
               '<rgb name="diffuseReflectance" value="%s"/><rgb name="specularReflectance" value="%s"/><boolean name="nonlinear" value="%s"/></bsdf>'
               % ("beckmann" if M.flags & 8 else "ggx", float(M.alpha), one, R, S, "true" if M.flags & 2 else "false"),
        }[t]
        twos = t == 1 or (M.flags & 1 and t not in (6, 7, 8))
        if M.flags & 4:
            inner = ('<bsdf type="twosided">%s</bsdf>' % (body % "")) if twos else body % ""
            out.append('\t<bsdf type="mask" id="mat%d"><rgb name="opacity" value="%s"/>%s</bsdf>' % (i, c(M.opacity), inner))
        elif twos:
            out.append('\t<bsdf type="twosided" id="mat%d">%s</bsdf>' % (i, body % ""))
        else:
            out.append('\t' + body % (' id="mat%d"' % i))
    tm, te = np.asarray(desc.tri_material), np.asarray(desc.tri_emitter)
    idx, pos = np.asarray(desc.indices), np.asarray(desc.positions)
    groups = sorted({(int(a), int(b)) for a, b in zip(tm, te)}, key=lambda g: (g[1] < 0, g[1], g[0]))  # emitters first, in emitter order
    # shapes in emitter order (the loader numbers emitters in shape order), shapes without an emitter last
    shapes = [("mesh", g) for g in groups] + [("sphere", sp) for sp in (getattr(desc, "spheres", None) or [])]
    em_of = lambda rec: rec[1][1] if rec[0] == "mesh" else int(rec[1].get("emitter", -1))  # noqa: E731
    shapes.sort(key=lambda rec: (em_of(rec) < 0, em_of(rec)))
    for gi, (kind, rec) in enumerate(shapes):
        if kind == "sphere":
            out.append('\t<shape type="sphere">')
            R = np.asarray(rec.get("to_world", np.eye(3)), np.float32).reshape(3, 3)
            if np.array_equal(R, np.eye(3, dtype=np.float32)):
                out.append('\t\t<point name="center" x="%r" y="%r" z="%r"/>' % tuple(float(np.float32(v)) for v in rec["center"]))
            else:
                M4 = np.eye(4, dtype=np.float32); M4[:3, :3] = R; M4[:3, 3] = rec["center"]
                out.append('\t\t<transform name="toWorld"><matrix value="%s"/></transform>' % " ".join(repr(float(x)) for x in M4.reshape(-1)))
            out.append('\t\t<float name="radius" value="%r"/>' % float(np.float32(rec["radius"])))
            if rec.get("flip_normals"):
                out.append('\t\t<boolean name="flipNormals" value="true"/>')
            out.append('\t\t<ref id="mat%d"/>' % int(rec.get("material", 0)))
            if int(rec.get("emitter", -1)) >= 0:
                out.append('\t\t<emitter type="area"><rgb name="radiance" value="%s"/></emitter>' % c(desc.emitters[int(rec["emitter"])]["radiance"]))
            out.append('\t</shape>')
            continue
        mat, em = rec
        sel = np.nonzero((tm == mat) & (te == em))[0]
        fn = "meshes/%s_%03d.obj" % (name, gi)
        with open(os.path.join(directory, fn), "w") as f:
            used = np.unique(idx[sel])
            remap = {int(v): k + 1 for k, v in enumerate(used)}
            for v in used:
                f.write("v %r %r %r\n" % tuple(float(x) for x in pos[v]))
            if desc.normals is not None:
                for v in used:
                    f.write("vn %r %r %r\n" % tuple(float(x) for x in np.asarray(desc.normals)[v]))
            for t in sel:
                f.write("f " + " ".join(("%d//%d" % (remap[int(v)], remap[int(v)])) if desc.normals is not None else str(remap[int(v)]) for v in idx[t]) + "\n")
        out.append('\t<shape type="obj">')
        out.append('\t\t<string name="filename" value="%s"/>' % fn)
        if desc.normals is None:
            out.append('\t\t<boolean name="faceNormals" value="true"/>')
        out.append('\t\t<ref id="mat%d"/>' % mat)
        if em >= 0:
            out.append('\t\t<emitter type="area"><rgb name="radiance" value="%s"/></emitter>' % c(desc.emitters[em]["radiance"]))
        out.append('\t</shape>')
    if getattr(desc, "environment", None) is not None:
        out.append('\t<emitter type="constant"><rgb name="radiance" value="%s"/></emitter>' % c(desc.environment))
    if getattr(desc, "envmap", None) is not None:
        from . import imageio
        imageio.write_pfm(os.path.join(directory, name + "_envmap.pfm"), desc.envmap["rgb"])
        M4 = np.eye(4, dtype=np.float32); M4[:3, :3] = np.asarray(desc.envmap.get("to_world", np.eye(3)), np.float32).reshape(3, 3)
        out.append('\t<emitter type="envmap"><string name="filename" value="%s_envmap.pfm"/><float name="scale" value="%r"/>'
                   '<transform name="toWorld"><matrix value="%s"/></transform></emitter>'
                   % (name, float(desc.envmap.get("scale", 1.0)), " ".join(repr(float(x)) for x in M4.reshape(-1))))
    out.append('</scene>')
    path = os.path.join(directory, name + ".xml")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return path
