#!/usr/bin/env python3
"""Dev-time tool: derive the linear-RGB values of the CBOX spectra the way Mitsuba's RGB build does.

Mitsuba (SPECTRUM_SAMPLES=3) turns <spectrum value="l:v, ..."> into RGB by
  * zeroExtend(): pad with a zero one average-spacing beyond each non-zero end (spectrum.cpp:630-648),
  * integrating spectrum x CIE 1931 matching functions over 360..830 nm, normalised by the integral
    of y-bar (spectrum.cpp:172-186),
  * InterpolatedSpectrum::eval() interpolating *backwards* inside each interval —
    lerp(t, fb, fa) instead of lerp(t, fa, fb) (spectrum.cpp:693-706).  With the emitter's 100 nm
    spacing this quirk lowers R by 11 %; it is reproduced here because the reference's shipped
    render (scenes/cbox/cbox.exr) only matches with it (R/G/B ratios 1.00/1.00/1.00 vs 1.125/0.99/1.005),
  * XYZ -> Rec.709 (spectrum.cpp:256-261), clampNegative.
The CIE tables are read from the reference checkout at dev time (they are not copied into this
repository); only the six resulting RGB triples are committed, as constants in ppg_host/scenes.py.
"""
import re
import xml.etree.ElementTree as ET
import numpy as np

REF = "/root/reference"


def table(src, name):
    m = re.search(r"const Float " + name + r"\[CIE_samples\] = \{(.*?)\};", src, re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    return np.array([float(x.rstrip("f")) for x in re.findall(r"[-+0-9.eE]+f?", body)])


def eval_interp(lam, val, g):
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


def zero_extend(lam, val):
    lam, val = list(lam), list(val)
    spacing = (lam[-1] - lam[0]) / (len(lam) - 1)
    if val[0] != 0:
        lam.insert(0, lam[0] - spacing); val.insert(0, 0.0)
    if val[-1] != 0:
        lam.append(lam[-1] + spacing); val.append(0.0)
    return lam, val


def main():
    src = open(REF + "/mitsuba/src/libcore/spectrum.cpp").read()
    wl, X, Y, Z = (table(src, n) for n in ("CIE_wavelengths", "CIE_X_entries", "CIE_Y_entries", "CIE_Z_entries"))
    assert len(wl) == 471 and wl[0] == 360 and wl[-1] == 830
    grid = np.arange(360.0, 830.0, 0.00731)  # off-node sampling: exact hits have measure zero
    cx, cy, cz = (eval_interp(wl, t, grid) for t in (X, Y, Z))
    ynorm = np.trapezoid(cy, grid)
    M = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])

    def to_rgb(spec):
        pts = [tuple(float(v) for v in tok.split(":")) for tok in spec.replace(" ", "").split(",")]
        lam, val = zero_extend(*zip(*pts))
        s = eval_interp(lam, val, grid)
        xyz = np.array([np.trapezoid(s * c, grid) for c in (cx, cy, cz)]) / ynorm
        return np.maximum(M @ xyz, 0.0)

    root = ET.parse(REF + "/scenes/cbox/cbox.xml").getroot()
    for bsdf in root.findall("bsdf"):
        print(bsdf.get("id"), ", ".join("%.9g" % v for v in np.float32(to_rgb(bsdf.find("spectrum").get("value")))))
    for shape in root.findall("shape"):
        em = shape.find("emitter")
        if em is not None:
            print("emitter", ", ".join("%.9g" % v for v in np.float32(to_rgb(em.find("spectrum").get("value")))))


if __name__ == "__main__":
    main()
