"""Image output of the render driver: OpenEXR (scanline, uncompressed, float32 R/G/B, string attributes such as the
render log that hdrfilm attaches — hdrfilm.cpp:481-537) and PFM.  struct + numpy only; `read_exr` reads back what
`write_exr` writes (and any uncompressed float scanline file) for tests."""
import struct

import numpy as np

_MAGIC = 20000630


def _attr(name, typ, payload):
    return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(payload)) + payload


def write_exr(path, rgb, attributes=None):
    """rgb: float array [H, W, 3]; attributes: {name: str} stored as EXR `string` attributes (e.g. "log")."""
    img = np.ascontiguousarray(rgb, np.float32)
    H, W, C = img.shape
    assert C == 3
    chans = b""
    for name in ("B", "G", "R"):  # channels are stored in alphabetical order
        chans += name.encode() + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1)  # FLOAT, pLinear 0, no subsampling
    chans += b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    hdr = struct.pack("<II", _MAGIC, 2)
    hdr += _attr("channels", "chlist", chans)
    hdr += _attr("compression", "compression", b"\0")
    hdr += _attr("dataWindow", "box2i", box) + _attr("displayWindow", "box2i", box)
    hdr += _attr("lineOrder", "lineOrder", b"\0")
    hdr += _attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    hdr += _attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
    hdr += _attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
    for k, v in (attributes or {}).items():
        hdr += _attr(k, "string", str(v).encode("utf-8", "replace"))
    hdr += b"\0"
    line_bytes = 3 * 4 * W
    first = len(hdr) + 8 * H
    offsets = struct.pack("<%dQ" % H, *[first + y * (8 + line_bytes) for y in range(H)])
    with open(path, "wb") as f:
        f.write(hdr); f.write(offsets)
        for y in range(H):
            f.write(struct.pack("<ii", y, line_bytes))
            f.write(img[y, :, 2].tobytes()); f.write(img[y, :, 1].tobytes()); f.write(img[y, :, 0].tobytes())


def read_exr(path):
    """→ (rgb float32 [H, W, 3], {string attribute: str}); uncompressed float scanline files only."""
    buf = open(path, "rb").read()
    assert struct.unpack_from("<I", buf, 0)[0] == _MAGIC, "not an EXR file"
    off, attrs = 8, {}
    while buf[off] != 0:
        e = buf.index(b"\0", off); name = buf[off:e].decode(); off = e + 1
        e = buf.index(b"\0", off); typ = buf[off:e].decode(); off = e + 1
        size = struct.unpack_from("<i", buf, off)[0]; off += 4
        attrs[name] = (typ, buf[off:off + size]); off += size
    off += 1
    assert attrs["compression"][1] == b"\0", "compressed EXR: use tools/exr_min.py"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    names, o, data = [], 0, attrs["channels"][1]
    while data[o] != 0:
        e = data.index(b"\0", o); names.append(data[o:e].decode()); o = e + 1
        assert struct.unpack_from("<i", data, o)[0] == 2, "float channels only"
        o += 16
    out = np.zeros((H, W, 3), np.float32)
    for bo in struct.unpack_from("<%dQ" % H, buf, off):
        y, size = struct.unpack_from("<ii", buf, bo)
        row = np.frombuffer(buf, np.float32, len(names) * W, bo + 8).reshape(len(names), W)
        for k, n in enumerate(names):
            if n in "RGB":
                out[y - y0, :, "RGB".index(n)] = row[k]
    return out, {k: v[1].decode("utf-8", "replace") for k, v in attrs.items() if v[0] == "string"}


def write_pfm(path, rgb):
    img = np.ascontiguousarray(rgb, np.float32)
    H, W, _ = img.shape
    with open(path, "wb") as f:
        f.write(("PF\n%d %d\n-1.0\n" % (W, H)).encode())
        f.write(img[::-1].astype("<f4").tobytes())  # PFM stores the bottom row first
