"""Generates tests/golden/rtrans_slices.npz: rough-transmittance slices (ppg_scene.rtrans layout) for the roughplastic materials the
tests use, cut from Mitsuba's data/microfacet/{ggx,beckmann}.dat exactly as RoughPlastic::configure() does (ppg_host/rtrans.py).

Run in the development container, where the reference tree is mounted:
    python tests/golden/make_rtrans_slices.py [/root/reference/mitsuba/data]
The tables are reference *data*; the slices are ~100 floats per material.  Also stored: a few raw table entries, to pin the file
parser (rtrans.h:81-146) independently of the reduction.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "practical-path-guiding_amd"))
from ppg_host import rtrans  # noqa: E402

CASES = [("ggx", 0.1, 1.5), ("ggx", 0.3, 1.49), ("beckmann", 0.2, 1.49), ("ggx", 0.05, 1.9), ("beckmann", 0.4, 1.33)]

if __name__ == "__main__":
    data = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/mitsuba/data"
    out = {"cases": np.array([(d, "%r" % a, "%r" % e) for d, a, e in CASES])}
    for i, (d, a, e) in enumerate(CASES):
        out["slice%d" % i] = rtrans.roughplastic_slice(d, a, e, data)
    for d in ("ggx", "beckmann"):
        t = rtrans.RoughTransmittance(os.path.join(data, "microfacet", d + ".dat"))
        out[d + "_shape"] = np.array([t.eta_samples, t.alpha_samples, t.theta_samples])
        out[d + "_range"] = np.array([t.eta_min, t.eta_max, t.alpha_min, t.alpha_max], np.float32)
        out[d + "_probe"] = np.concatenate([t.trans[7, 11, ::9], t.trans[50 + 3, 40, ::9], t.diff[::17, 5]]).astype(np.float32)
    np.savez(os.path.join(HERE, "rtrans_slices.npz"), **out)
    print("wrote", os.path.join(HERE, "rtrans_slices.npz"), {k: v.shape for k, v in out.items()})
