#!/usr/bin/env python3
"""Dev-time tool: write the CIE 1931 2-degree standard-observer colour matching functions (360..830 nm, 1 nm;
the public CIE table, as tabulated in the reference's spectrum.cpp) to ppg_host/data/cie1931_xyz_1nm.npy so that
the Mitsuba-XML loader can convert <spectrum value="l:v, ..."> on machines without the reference checkout.
Data only: 471 x 3 numbers."""
import os
import re

import numpy as np

REF = "/root/reference/mitsuba/src/libcore/spectrum.cpp"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "practical-path-guiding_amd", "ppg_host", "data", "cie1931_xyz_1nm.npy")


def table(src, name):
    m = re.search(r"const Float " + name + r"\[CIE_samples\] = \{(.*?)\};", src, re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    return np.array([float(x.rstrip("f")) for x in re.findall(r"[-+0-9.eE]+f?", body)])


if __name__ == "__main__":
    src = open(REF).read()
    wl = table(src, "CIE_wavelengths")
    assert len(wl) == 471 and wl[0] == 360 and wl[-1] == 830 and np.all(np.diff(wl) == 1)
    xyz = np.stack([table(src, n) for n in ("CIE_X_entries", "CIE_Y_entries", "CIE_Z_entries")], 1)
    np.save(OUT, xyz.astype(np.float64))
    inc = os.path.join(os.path.dirname(os.path.dirname(OUT)), "..", "host", "cie1931_xyz_1nm.inc")
    with open(inc, "w") as f:  # the same numbers for the C++ scene loader (host/scene_xml.h)
        f.write("// CIE 1931 2-degree standard observer, 360..830 nm in 1 nm steps: x-bar, y-bar, z-bar (data; written by tools/make_cie_table.py\n"
                "// from the same table as ppg_host/data/cie1931_xyz_1nm.npy)\n")
        for row in xyz:
            f.write("{%r, %r, %r},\n" % tuple(float(v) for v in row))
    print(OUT, xyz.shape, xyz.sum(0))
