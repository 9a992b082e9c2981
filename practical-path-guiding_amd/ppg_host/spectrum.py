"""Spectrum values of a Mitsuba scene file → the linear RGB triple the reference's RGB build renders with.

Mitsuba compiled with SPECTRUM_SAMPLES = 3 (config-linux-gcc.py:7) turns
  <rgb value="r, g, b"/>, <rgb value="#rrggbb"/>      into the triple itself            (scenehandler.cpp, "rgb")
  <srgb value=.../>                                   into the sRGB-decoded triple      (spectrum.cpp fromSRGB)
  <spectrum value="v"/>                               into (v, v, v)
  <spectrum value="l0:v0, l1:v1, ..."/>               into RGB by
      * zeroExtend(): pad with a zero one average-spacing beyond each non-zero end (spectrum.cpp:630-648),
      * integrating spectrum x CIE 1931 matching functions over 360..830 nm, normalised by the integral of
        y-bar (spectrum.cpp:172-186),
      * InterpolatedSpectrum::eval() interpolating *backwards* inside each interval — lerp(t, fb, fa) instead of
        lerp(t, fa, fb) (spectrum.cpp:693-706).  With the CBOX emitter's 100 nm spacing this quirk lowers R by
        11 %; it is reproduced because the reference's shipped render (scenes/cbox/cbox.exr) only matches with it,
      * XYZ -> Rec.709 (spectrum.cpp:256-261), clampNegative.
The reference integrates adaptively (Gauss-Lobatto, spectrum.cpp:546-568); a fine off-node trapezoid grid is
used here (the integrand is piecewise quadratic).  The CIE table is data/cie1931_xyz_1nm.npy (tools/make_cie_table.py).
"""
import os

import numpy as np

_XYZ_TO_RGB = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
_cache = {}


def _eval_interp(lam, val, g):
    """InterpolatedSpectrum::eval incl. the reversed lerp; exact hits return the table value."""
    lam, val = np.asarray(lam, float), np.asarray(val, float)
    i = np.searchsorted(lam, g, side="left")
    ic = np.clip(i, 0, len(lam) - 1)
    inside = (g >= lam[0]) & (g <= lam[-1])
    exact = inside & (lam[ic] == g)
    ii = np.clip(i, 1, len(lam) - 1)
    a, b, fa, fb = lam[ii - 1], lam[ii], val[ii - 1], val[ii]
    t = (g - a) / (b - a)
    out = np.where(inside, (1 - t) * fb + t * fa, 0.0)
    out[exact] = val[ic][exact]
    return out


def _zero_extend(lam, val):
    lam, val = list(lam), list(val)
    spacing = (lam[-1] - lam[0]) / (len(lam) - 1)
    if val[0] != 0:
        lam.insert(0, lam[0] - spacing); val.insert(0, 0.0)
    if val[-1] != 0:
        lam.append(lam[-1] + spacing); val.append(0.0)
    return lam, val


def _cie():
    if "cie" not in _cache:
        xyz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "cie1931_xyz_1nm.npy"))
        wl = np.arange(360.0, 831.0)
        grid = np.arange(360.0, 830.0, 0.00731)  # off-node sampling: exact hits have measure zero
        c = [_eval_interp(wl, xyz[:, k], grid) for k in range(3)]
        _cache["cie"] = (grid, c, np.trapezoid(c[1], grid))
    return _cache["cie"]


def interpolated_to_rgb(pairs, zero_extend=True, clamp=True):
    """[(lambda_nm, value), ...] → float32 RGB (Spectrum::fromContinuousSpectrum of the RGB build).  Scene-file spectra are zero-extended and
    clamped (scenehandler.cpp:563-566); the conductor plug-ins read data/ior/*.spd without either (roughconductor.cpp:179-183)."""
    grid, c, ynorm = _cie()
    lam, val = _zero_extend(*zip(*pairs)) if zero_extend else zip(*pairs)
    s = _eval_interp(lam, val, grid)
    xyz = np.array([np.trapezoid(s * ck, grid) for ck in c]) / ynorm
    rgb = _XYZ_TO_RGB @ xyz
    return (np.maximum(rgb, 0.0) if clamp else rgb).astype(np.float32)


def read_spd(path):
    """InterpolatedSpectrum(path), spectrum.cpp:575-600: "wavelength value" lines, # comments; stops at the first malformed line."""
    pairs = []
    for line in open(path, errors="replace"):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        tok = line.split()
        try:
            pairs.append((float(tok[0]), float(tok[1])))
        except (ValueError, IndexError):
            break
    return pairs


def blackbody_to_rgb(temperature, scale=1.0):
    """<blackbody temperature="5000K" scale=".."/> (scenehandler.cpp:534-547): Planck's law in W m^-2 nm^-1 sr^-1 (BlackBodySpectrum::eval,
    spectrum.cpp:483-495) through Spectrum::fromContinuousSpectrum like any other continuous spectrum, clamped, times `scale`."""
    grid, c, ynorm = _cie()
    lam = grid * 1e-9
    cc, k, h = 299792458.0, 1.3806488e-23, 6.62606957e-34
    s = (2 * h * cc * cc) * np.power(lam, -5.0) / ((np.exp((h / k) * cc / (lam * float(temperature))) - 1.0) * 1e9)
    xyz = np.array([np.trapezoid(s * ck, grid) for ck in c]) / ynorm
    return (np.maximum(_XYZ_TO_RGB @ xyz, 0.0).astype(np.float32) * np.float32(scale)).astype(np.float32)


def srgb_to_linear(v):
    v = np.asarray(v, np.float64)
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4).astype(np.float32)


def _triple(text):
    text = text.strip()
    if text.startswith("#"):
        h = text[1:]
        return [int(h[k:k + 2], 16) / 255.0 for k in (0, 2, 4)]
    v = [float(t) for t in text.replace(",", " ").split()]
    if len(v) == 1:
        v = v * 3
    if len(v) != 3:
        raise ValueError("expected 1 or 3 colour components, got %r" % text)
    return v


def parse(tag, value):
    """tag in {"rgb", "srgb", "spectrum"}; value = the element's `value` attribute → float32 RGB."""
    if tag == "rgb":
        return np.asarray(_triple(value), np.float32)
    if tag == "srgb":
        return srgb_to_linear(_triple(value))
    if tag == "spectrum":
        if ":" in value:
            pairs = [tuple(float(v) for v in tok.split(":")) for tok in value.replace(" ", "").split(",") if tok]
            return interpolated_to_rgb(pairs)
        v = [float(t) for t in value.replace(",", " ").split()]
        if len(v) == 1:
            return np.full(3, v[0], np.float32)
        if len(v) == 3:  # SPECTRUM_SAMPLES values given directly
            return np.asarray(v, np.float32)
    raise ValueError("unsupported colour element <%s value=%r>" % (tag, value))
