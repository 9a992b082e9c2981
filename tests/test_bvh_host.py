"""The quantised BVH4 of the traversal kernels, checked on the CPU (no GPU needed).

`ppg_debug_build_bvh` (include/ppg_testhooks.h) runs the builder of `ppg_set_scene` on a triangle soup.  The tests restate the kernels' node test
(csrc/ppg_device.h `bvh4_children`: the RAY is scaled to the node's power-of-two grid, t = (q - o') * (1/d'), near / far plane by the
sign of 1/d, exit distance widened by 1 + 2 gamma_3) in numpy float32, operation for operation, and assert the property the closest
hit rests on: on the way from the root to a triangle that a ray really hits, no child box is ever reported as missed.  Also the
builder's own invariants: every triangle in exactly one leaf, every box contains what lies below it, cell exponents within the
range the kernels' scalings need."""
import ctypes as C

import numpy as np
import pytest

EMPTY = 0x7FFFFFFF
f32 = np.float32

NODE = np.dtype([("org", "<f4", 3), ("exps", "<u4"), ("qlo", "<u4", 3), ("qhi", "<u4", 3), ("child", "<i4", 4), ("pad", "<i4", 2)])
assert NODE.itemsize == 64


def build(lib_path, pos, idx, pad, max_leaf=3):
    lib = C.CDLL(lib_path)
    lib.ppg_debug_build_bvh.restype = C.c_int
    lib.ppg_debug_build_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_int32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    pos = np.ascontiguousarray(pos, np.float32)
    idx = np.ascontiguousarray(idx, np.uint32)
    n = idx.shape[0]
    cap = 2 * n + 8
    nodes = np.zeros(cap, NODE)
    order = np.zeros(n, np.uint32)
    nn = C.c_uint32(0)
    rc = lib.ppg_debug_build_bvh(pos.ctypes.data, idx.ctypes.data, n, float(pad), max_leaf, nodes.ctypes.data, cap, C.byref(nn), order.ctypes.data)
    assert rc == 0 and 0 < nn.value <= cap
    return nodes[: nn.value], order


def soup(rng, n, offset=0.0, flat=False, scale=1.0):
    c = rng.uniform(-1, 1, (n, 3)) * scale
    if flat:
        c[:, 2] = 0.25 * scale  # an axis-aligned sheet: boxes of zero extent along z
    e = rng.normal(0, 0.15 * scale, (n, 3, 3))
    if flat:
        e[:, :, 2] = 0
    v = (c[:, None, :] + e + offset).astype(np.float32)
    return v.reshape(-1, 3), np.arange(3 * n, dtype=np.uint32).reshape(n, 3)


def leaf_paths(nodes):
    """leaf-order triangle -> [(node, slot), ...] from the root down to the leaf slot that holds it"""
    paths = {}
    stack = [(0, [])]
    while stack:
        ni, path = stack.pop()
        for k in range(4):
            ch = int(nodes["child"][ni][k])
            if ch == EMPTY:
                continue
            here = path + [(ni, k)]
            if ch >= 0:
                stack.append((ch, here))
            else:
                code = ~ch
                first, cnt = code >> 3, (code & 7) + 1
                for q in range(first, first + cnt):
                    assert q not in paths, "triangle in two leaves"
                    paths[q] = here
    return paths


def scale_of(e):
    return np.array([int(e) << 23], np.uint32).view(np.float32)[0]


def child_hit(node, k, o, inv_d, mint, tlim):
    """bvh4_children for one child, in float32, operation for operation"""
    n_ = f32(mint)
    f_ = f32(np.inf)
    for a in range(3):
        e = (int(node["exps"]) >> (8 * a)) & 255
        s, inv_s = scale_of(e), scale_of(254 - e)
        os_ = f32(f32(o[a] - node["org"][a]) * inv_s)
        ids = f32(inv_d[a] * s)
        lo = f32((int(node["qlo"][a]) >> (8 * k)) & 255)
        hi = f32((int(node["qhi"][a]) >> (8 * k)) & 255)
        near, far = (hi, lo) if inv_d[a] < 0 else (lo, hi)
        tn = f32(f32(near - os_) * ids)
        tf = f32(f32(far - os_) * ids)
        n_ = max(n_, tn)
        f_ = min(f_, tf)
    f_ = min(f32(f_ * f32(1.0000008)), f32(tlim))
    return bool(n_ <= f_)


def safe_inv(d):
    with np.errstate(divide="ignore"):
        return np.where(d == 0, f32(1e30), f32(1) / d).astype(np.float32)


def robust_hits(pos, idx_in_leaf_order, o, d):
    """(ray, triangle, t) of the intersections well inside a triangle, in float64 (Moeller-Trumbore)"""
    v = pos[idx_in_leaf_order].astype(np.float64)  # [T, 3, 3]
    e1, e2 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    out = []
    for r in range(o.shape[0]):
        oo, dd = o[r].astype(np.float64), d[r].astype(np.float64)
        p = np.cross(dd, e2)
        det = np.einsum("ij,ij->i", e1, p)
        ok = np.abs(det) > 1e-12
        inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
        tv = oo - v[:, 0]
        u = np.einsum("ij,ij->i", tv, p) * inv
        q = np.cross(tv, e1)
        w = np.einsum("j,ij->i", dd, q) * inv
        t = np.einsum("ij,ij->i", e2, q) * inv
        good = ok & (u > 1e-3) & (w > 1e-3) & (u + w < 1 - 1e-3) & (t > 1e-3)
        for tri in np.nonzero(good)[0]:
            out.append((r, int(tri), float(t[tri])))
    return out


CASES = [
    dict(name="soup", n=1500, offset=0.0, flat=False, scale=1.0),
    dict(name="far-from-origin", n=800, offset=1000.0, flat=False, scale=1.0),
    dict(name="flat-sheet", n=800, offset=0.0, flat=True, scale=1.0),
    dict(name="tiny", n=600, offset=0.0, flat=False, scale=1e-4),
    dict(name="large", n=600, offset=0.0, flat=False, scale=1e5),
]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
@pytest.mark.parametrize("max_leaf", [3, 4])
def test_builder_invariants_and_conservative_node_test(hip_lib_path, case, max_leaf):
    rng = np.random.default_rng(1234 + case["n"])
    pos, idx = soup(rng, case["n"], case["offset"], case["flat"], case["scale"])
    ext = float(np.max(pos.max(0) - pos.min(0)))
    pad = 2e-6 * ext + 1e-30  # ppg_set_scene's padding
    nodes, order = build(hip_lib_path, pos, idx, pad, max_leaf)
    assert sorted(order.tolist()) == list(range(case["n"]))
    paths = leaf_paths(nodes)
    assert sorted(paths) == list(range(case["n"]))  # every triangle in exactly one leaf
    tri = idx[order]  # leaf order
    # exponents: what bvh4_children's exact scalings need (BvhBuilder::quantise)
    ex = np.stack([(nodes["exps"] >> (8 * a)) & 255 for a in range(3)], 1)
    assert ex.min() >= 64 and ex.max() <= 154
    # every child box (exact planes origin + q * cell, in double) contains the padded triangles of its leaf slots
    # (the builder pads in float32: lo = fl(min - pad), hi = fl(max + pad))
    lo_all = (pos[tri].min(1) - f32(pad)).astype(np.float32).astype(np.float64)
    hi_all = (pos[tri].max(1) + f32(pad)).astype(np.float32).astype(np.float64)
    for q, path in paths.items():
        for ni, k in path:
            nd = nodes[ni]
            for a in range(3):
                s = float(scale_of((int(nd["exps"]) >> (8 * a)) & 255))
                plo = float(nd["org"][a]) + ((int(nd["qlo"][a]) >> (8 * k)) & 255) * s
                phi = float(nd["org"][a]) + ((int(nd["qhi"][a]) >> (8 * k)) & 255) * s
                assert plo <= lo_all[q][a] and phi >= hi_all[q][a], (ni, k, a)
    # rays: from surface points (like path vertices), from far outside, and with exactly axis-parallel directions
    R = 300
    centre = pos.reshape(-1, 3).mean(0)
    o = np.empty((R, 3), np.float32)
    d = rng.normal(size=(R, 3))
    on_surface = pos[tri[rng.integers(0, case["n"], R)]].mean(1)
    o[: R // 2] = on_surface[: R // 2]
    o[R // 2:] = centre + rng.normal(size=(R - R // 2, 3)) * 4 * ext
    d[R // 2:] = centre - o[R // 2:] + rng.normal(size=(R - R // 2, 3)) * 0.3 * ext  # aimed at the scene
    d[::7, 0] = 0.0
    d[::11, 1] = 0.0
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d.astype(np.float32)
    hits = robust_hits(pos, tri, o, d)
    assert len(hits) > 100
    inv = safe_inv(d)
    eps = f32(1e-4)  # Epsilon of the ray (scaled by the origin's magnitude like the kernels' adaptive epsilon)
    missed = []
    for r, q, t in hits:
        mint = f32(eps * max(np.abs(o[r]).max(), eps))
        if t <= float(mint) * 4:
            continue
        tlim = f32(t * (1 + 1e-5))  # the best hit so far is never closer than the triangle itself
        for ni, k in paths[q]:
            if not child_hit(nodes[ni], k, o[r], inv[r], mint, tlim):
                missed.append((r, q, ni, k))
    assert not missed, missed[:5]


def test_single_triangle_and_degenerate_input(hip_lib_path):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    nodes, order = build(hip_lib_path, pos, np.array([[0, 1, 2]], np.uint32), 1e-6)
    assert order.tolist() == [0] and sorted(leaf_paths(nodes)) == [0]
    # coincident vertices (zero-area triangles) still end up in leaves
    pos = np.zeros((9, 3), np.float32)
    nodes, order = build(hip_lib_path, pos, np.arange(9, dtype=np.uint32).reshape(3, 3), 1e-30)
    assert sorted(leaf_paths(nodes)) == [0, 1, 2]
