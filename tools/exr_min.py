"""Minimal OpenEXR reader (header attributes + scanline NONE/ZIPS/ZIP, half/float channels).

Dev-time tool: used to mine the reference's shipped renders (scenes/*/*.exr) for their embedded
`log` attribute and pixel statistics.  zlib + numpy only.
"""
import struct, zlib
import numpy as np


def _cstr(buf, off):
    end = buf.index(b"\0", off)
    return buf[off:end].decode("latin1"), end + 1


def read_header(buf):
    assert struct.unpack_from("<I", buf, 0)[0] == 20000630, "not an EXR file"
    off = 8
    attrs = {}
    while buf[off] != 0:
        name, off = _cstr(buf, off)
        typ, off = _cstr(buf, off)
        size = struct.unpack_from("<i", buf, off)[0]
        off += 4
        attrs[name] = (typ, buf[off:off + size])
        off += size
    return attrs, off + 1


def parse_channels(data):
    chans, off = [], 0
    while data[off] != 0:
        name, off = _cstr(data, off)
        ptype, _plin, xs, ys = struct.unpack_from("<iB3xii", data, off)
        off += 16
        chans.append((name, ptype, xs, ys))
    return chans


def read_exr(path):
    """Returns (attrs, {channel_name: float32 array [H, W]})."""
    buf = open(path, "rb").read()
    attrs, off = read_header(buf)
    chans = parse_channels(attrs["channels"][1])
    comp = attrs["compression"][1][0]
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    lines_per_block = {0: 1, 2: 1, 3: 16}[comp]
    nblocks = (H + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from("<%dQ" % nblocks, buf, off)
    bpp = {1: 2, 2: 4}
    out = {c[0]: np.zeros((H, W), np.float32) for c in chans}
    for bo in offsets:
        y, size = struct.unpack_from("<ii", buf, bo)
        raw = buf[bo + 8: bo + 8 + size]
        nl = min(lines_per_block, y1 - y + 1)
        expect = nl * sum(bpp[c[1]] * W for c in chans)
        if comp != 0 and size < expect:
            d = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int32)
            # predictor + interleave undo
            d = (np.cumsum(np.concatenate([[d[0]], d[1:] - 128])) & 255).astype(np.uint8)
            half = (len(d) + 1) // 2
            r = np.empty(len(d), np.uint8)
            r[0::2] = d[:half]
            r[1::2] = d[half:]
            raw = r.tobytes()
        p = 0
        for ly in range(nl):
            for name, ptype, _, _ in chans:
                n = bpp[ptype] * W
                arr = np.frombuffer(raw, np.float16 if ptype == 1 else np.float32, W, p)
                out[name][y - y0 + ly] = arr.astype(np.float32)
                p += n
    return attrs, out


def attr_string(attrs, name):
    typ, data = attrs[name]
    return data.decode("latin1") if typ == "string" else None
