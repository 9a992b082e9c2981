"""Rough-transmittance slices for the `roughplastic` BSDF (include/ppg.h: ppg_scene.rtrans).

Mitsuba keeps the transmittance through a rough dielectric boundary in precomputed tables, data/microfacet/{beckmann,ggx}.dat, and
RoughPlastic::configure() (roughplastic.cpp:285-305) cuts them down to what one material needs: a 1D curve over the warped
incident cosine (setEta, setAlpha) and one number for the diffuse transmittance from the inside.  This module restates that
reduction — the file format of rtrans.h:81-146, setEta / setAlpha / evalDiffuse of rtrans.h:233-400 and the cubic interpolation
of spline.cpp:23-60, 236-452 — in float32, so the host hands the integrator the same slice the plug-in would have used.  The
tables themselves are Mitsuba data: they are read from the operator's Mitsuba tree (`data_dir`, or $PPG_MITSUBA_DATA), never bundled.
"""
import os

import numpy as np

f32 = np.float32
HEADER = b"MTS_TRANSMITTANCE"


def _powf(x, y):
    """std::pow(float, float) of the C library — numpy's float32 power is its own SIMD routine and differs in the last bit."""
    global _libm
    try:
        _libm
    except NameError:
        import ctypes
        import ctypes.util
        _libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _libm.powf.restype, _libm.powf.argtypes = ctypes.c_float, [ctypes.c_float, ctypes.c_float]
    return f32(_libm.powf(float(f32(x)), float(f32(y))))


class RoughTransmittanceError(Exception):
    pass


def _weights(p, size):
    """Per-dimension knot and node weights of evalCubicInterp2D/3D (spline.cpp:244-288, 387-431), vectorised over `p` (float32 in
    [0, 1]); returns (knot[int], weights[4, n])."""
    p = np.asarray(p, f32)
    if not np.all((p >= 0) & (p <= 1)):
        raise RoughTransmittanceError("interpolation argument outside [0, 1]")
    t = (p * f32(size - 1)) / f32(1)
    knot = np.minimum(t.astype(np.int64), size - 2)
    t = t - knot.astype(f32)
    t2 = t * t
    t3 = t2 * t
    w0 = np.zeros_like(t)
    w1 = f32(2) * t3 - f32(3) * t2 + f32(1)
    w2 = f32(-2) * t3 + f32(3) * t2
    w3 = np.zeros_like(t)
    d0 = t3 - f32(2) * t2 + t
    d1 = t3 - t2
    lo = knot > 0
    w2 = w2 + np.where(lo, f32(0.5) * d0, d0)
    w0 = w0 - np.where(lo, f32(0.5) * d0, f32(0))
    w1 = w1 - np.where(lo, f32(0), d0)
    hi = knot + 2 < size
    w3 = w3 + np.where(hi, f32(0.5) * d1, f32(0))
    w1 = w1 - np.where(hi, f32(0.5) * d1, d1)
    w2 = w2 + np.where(hi, f32(0), d1)
    return knot, np.stack([w0, w1, w2, w3]).astype(f32)


def cubic_interp_1d(x, values):
    """evalCubicInterp1D(x, values, size, 0, 1) (spline.cpp:23-60)."""
    values = np.asarray(values, f32)
    size = len(values)
    x = f32(x)
    if not (x >= 0 and x <= 1):
        return f32(0)
    t = (x * f32(size - 1)) / f32(1)
    k = max(0, min(int(t), size - 2))
    f0, f1 = values[k], values[k + 1]
    d0 = f32(0.5) * (values[k + 1] - values[k - 1]) if k > 0 else values[k + 1] - values[k]
    d1 = f32(0.5) * (values[k + 2] - values[k]) if k + 2 < size else values[k + 1] - values[k]
    t = t - f32(k)
    t2 = t * t
    t3 = t2 * t
    return ((f32(2) * t3 - f32(3) * t2 + f32(1)) * f0 + (f32(-2) * t3 + f32(3) * t2) * f1 + (t3 - f32(2) * t2 + t) * d0 + (t3 - t2) * d1)


def cubic_interp_nd(points, values):
    """evalCubicInterp2D / 3D: `values` is indexed [z][y][x] (x fastest), `points` is a list of per-dimension coordinate arrays in the
    order (x, y[, z]) that broadcast against each other.  Terms are added in the reference's loop order (z, y, x ascending)."""
    values = np.asarray(values, f32)
    dims = len(points)
    sizes = values.shape[::-1]
    pts = np.broadcast_arrays(*[np.asarray(p, f32) for p in points])
    shape = pts[0].shape
    kw = [_weights(p.reshape(-1), sizes[d]) for d, p in enumerate(pts)]
    result = np.zeros(pts[0].size, f32)
    if dims == 2:
        (kx, wx), (ky, wy) = kw
        for y in range(-1, 3):
            for x in range(-1, 3):
                w = wx[x + 1] * wy[y + 1]
                v = values[np.clip(ky + y, 0, sizes[1] - 1), np.clip(kx + x, 0, sizes[0] - 1)]
                result = result + np.where(w == 0, f32(0), v * w)
    else:
        (kx, wx), (ky, wy), (kz, wz) = kw
        for z in range(-1, 3):
            for y in range(-1, 3):
                wyz = wy[y + 1] * wz[z + 1]
                for x in range(-1, 3):
                    w = wx[x + 1] * wyz
                    v = values[np.clip(kz + z, 0, sizes[2] - 1), np.clip(ky + y, 0, sizes[1] - 1), np.clip(kx + x, 0, sizes[0] - 1)]
                    result = result + np.where(w == 0, f32(0), v * w)
    return result.reshape(shape)


class RoughTransmittance:
    """rtrans.h's RoughTransmittance: the table of one microfacet distribution, reducible to a fixed eta and then a fixed alpha."""

    def __init__(self, path):
        raw = open(path, "rb").read()
        if raw[:len(HEADER)] != HEADER:
            raise RoughTransmittanceError("%s: not a rough-transmittance data file" % path)
        o = len(HEADER)
        self.eta_samples, self.alpha_samples, self.theta_samples = (int(v) for v in np.frombuffer(raw, "<u8", 3, o))
        o += 24
        self.eta_min, self.eta_max, self.alpha_min, self.alpha_max = (f32(v) for v in np.frombuffer(raw, "<f4", 4, o))
        o += 16
        n = 2 * self.eta_samples * self.alpha_samples * (self.theta_samples + 1)
        if len(raw) != o + 4 * n:
            raise RoughTransmittanceError("%s: truncated rough-transmittance data file" % path)
        data = np.frombuffer(raw, "<f4", n, o).reshape(2 * self.eta_samples, self.alpha_samples, self.theta_samples + 1)
        self.trans = np.ascontiguousarray(data[..., :-1], f32)   # [2 * eta][alpha][theta]
        self.diff = np.ascontiguousarray(data[..., -1], f32)     # [2 * eta][alpha]
        self.eta_fixed = self.alpha_fixed = False

    def check(self, alpha, eta):  # checkAlpha / checkEta, rtrans.h:402-420
        alpha, eta = f32(alpha), f32(eta)
        if eta < 1:
            eta = f32(1) / eta
        if eta < self.eta_min or eta > self.eta_max:
            raise RoughTransmittanceError("relative IOR %g is outside the tabulated range [%g, %g]" % (eta, self.eta_min, self.eta_max))
        if alpha < self.alpha_min or alpha > self.alpha_max:
            raise RoughTransmittanceError("roughness alpha = %g is outside the tabulated range [%g, %g]" % (alpha, self.alpha_min, self.alpha_max))

    def _warp(self, v, lo, hi):
        return _powf((f32(v) - lo) / (hi - lo), 0.25)

    def set_eta(self, eta):  # rtrans.h:299-349
        assert not self.eta_fixed
        eta = f32(eta)
        trans, diff = self.trans[:self.eta_samples], self.diff[:self.eta_samples]
        if eta < 1:
            trans, diff = self.trans[self.eta_samples:], self.diff[self.eta_samples:]
            eta = f32(1) / eta
        if eta < self.eta_min:
            eta = self.eta_min
        w_eta = self._warp(eta, self.eta_min, self.eta_max)
        d_alpha, d_theta = f32(1) / f32(self.alpha_samples - 1), f32(1) / f32(self.theta_samples - 1)
        i = np.arange(self.alpha_samples).astype(f32) * d_alpha
        j = np.arange(self.theta_samples).astype(f32) * d_theta
        # i * dAlpha can exceed 1 by an ulp at the last knot in float32; the reference has the same argument and then returns 0 there
        out = RoughTransmittance.__new__(RoughTransmittance)
        out.__dict__.update(self.__dict__)
        ok_i, ok_j = i <= 1, j <= 1
        t = np.zeros((self.alpha_samples, self.theta_samples), f32)
        t[np.ix_(ok_i, ok_j)] = cubic_interp_nd([j[ok_j][None, :], i[ok_i][:, None], w_eta], trans)
        d = np.zeros(self.alpha_samples, f32)
        d[ok_i] = cubic_interp_nd([i[ok_i], w_eta], diff)
        out.trans, out.diff, out.eta_fixed = t, d, True
        return out

    def set_alpha(self, alpha):  # rtrans.h:357-400
        assert self.eta_fixed and not self.alpha_fixed
        w_alpha = self._warp(alpha, self.alpha_min, self.alpha_max)
        d_theta = f32(1) / f32(self.theta_samples - 1)
        j = np.arange(self.theta_samples).astype(f32) * d_theta
        ok = j <= 1
        out = RoughTransmittance.__new__(RoughTransmittance)
        out.__dict__.update(self.__dict__)
        t = np.zeros(self.theta_samples, f32)
        t[ok] = cubic_interp_nd([j[ok], w_alpha], self.trans)
        out.trans = t
        out.diff = np.array([cubic_interp_1d(w_alpha, self.diff)], f32)
        out.alpha_fixed = True
        return out

    def eval_diffuse(self, alpha):  # rtrans.h:249-258 (eta fixed)
        assert self.eta_fixed
        result = self.diff[0] if self.alpha_fixed else cubic_interp_1d(self._warp(alpha, self.alpha_min, self.alpha_max), self.diff)
        return min(f32(1), max(f32(0), f32(result)))

    def eval(self, cos_theta):  # rtrans.h:185-196, 233 (eta and alpha fixed)
        assert self.eta_fixed and self.alpha_fixed
        if not cos_theta >= 0:
            return f32(0)
        return min(f32(1), max(f32(0), cubic_interp_1d(_powf(abs(cos_theta), 0.25), self.trans)))


def find_data_dir(data_dir=None):
    data_dir = data_dir or os.environ.get("PPG_MITSUBA_DATA")
    if not data_dir:
        raise RoughTransmittanceError("roughplastic needs Mitsuba's data/microfacet tables: pass data_dir (--data-dir) or set PPG_MITSUBA_DATA "
                                      "to the `data` directory of a Mitsuba tree")
    return data_dir


_tables = {}


def roughplastic_slice(distribution, alpha, eta, data_dir=None):
    """What RoughPlastic::configure() precomputes for (distribution, alpha, eta), in ppg_scene.rtrans layout: theta_samples values of the
    external transmittance, then the internal diffuse transmittance."""
    path = os.path.join(find_data_dir(data_dir), "microfacet", "%s.dat" % distribution)
    if path not in _tables:
        if not os.path.exists(path):
            raise RoughTransmittanceError("%s not found (data/microfacet of a Mitsuba tree)" % path)
        _tables[path] = RoughTransmittance(path)
    table = _tables[path]
    alpha, eta = f32(max(f32(alpha), f32(1e-4))), f32(eta)  # MicrofacetDistribution clamps alpha (microfacet.h:135)
    table.check(alpha, eta)
    ext = table.set_eta(eta).set_alpha(alpha)
    internal = table.set_eta(f32(1) / eta)
    return np.concatenate([ext.trans, [internal.eval_diffuse(alpha)]]).astype(f32)
