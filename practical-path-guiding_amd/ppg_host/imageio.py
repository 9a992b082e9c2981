"""Image output of the render driver: OpenEXR (scanline, uncompressed, float32 R/G/B, string attributes such as the
render log that hdrfilm attaches — hdrfilm.cpp:481-537) and PFM.  struct + numpy only; `read_exr` / `read_pfm` / `read_hdr` read environment maps
(scanline EXR with NONE / ZIP compression, PFM, Radiance RGBE)."""
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
    """→ (rgb float32 [H, W, 3], {string attribute: str}); scanline files, compression NONE / ZIPS / ZIP, half or float channels."""
    import zlib
    buf = open(path, "rb").read()
    assert struct.unpack_from("<I", buf, 0)[0] == _MAGIC, "not an EXR file"
    off, attrs = 8, {}
    while buf[off] != 0:
        e = buf.index(b"\0", off); name = buf[off:e].decode(); off = e + 1
        e = buf.index(b"\0", off); typ = buf[off:e].decode(); off = e + 1
        size = struct.unpack_from("<i", buf, off)[0]; off += 4
        attrs[name] = (typ, buf[off:off + size]); off += size
    off += 1
    comp = attrs["compression"][1][0]
    if comp not in (0, 2, 3):
        raise ValueError("%s: EXR compression %d is not supported (NONE, ZIPS, ZIP)" % (path, comp))
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    W, H = x1 - x0 + 1, y1 - y0 + 1
    chans, o, data = [], 0, attrs["channels"][1]
    while data[o] != 0:
        e = data.index(b"\0", o); cname = data[o:e].decode(); o = e + 1
        ptype = struct.unpack_from("<i", data, o)[0]
        if ptype not in (1, 2):
            raise ValueError("%s: channel %s is neither half nor float" % (path, cname))
        chans.append((cname, ptype)); o += 16
    bpp = {1: 2, 2: 4}
    lines = {0: 1, 2: 1, 3: 16}[comp]
    out = np.zeros((H, W, 3), np.float32)
    for bo in struct.unpack_from("<%dQ" % ((H + lines - 1) // lines), buf, off):
        y, size = struct.unpack_from("<ii", buf, bo)
        raw = buf[bo + 8:bo + 8 + size]
        nl = min(lines, y1 - y + 1)
        if comp != 0 and size < nl * sum(bpp[t] * W for _, t in chans):
            d = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int32)
            d = (np.cumsum(np.concatenate([[d[0]], d[1:] - 128])) & 255).astype(np.uint8)  # undo the predictor ...
            half = (len(d) + 1) // 2
            r = np.empty(len(d), np.uint8)
            r[0::2] = d[:half]; r[1::2] = d[half:]                                           # ... and the byte interleave
            raw = r.tobytes()
        p = 0
        for ly in range(nl):
            for cname, ptype in chans:
                arr = np.frombuffer(raw, np.float16 if ptype == 1 else np.float32, W, p)
                p += bpp[ptype] * W
                if cname in ("R", "G", "B"):
                    out[y - y0 + ly, :, "RGB".index(cname)] = arr
                elif cname == "Y":
                    out[y - y0 + ly] = arr.astype(np.float32)[:, None]
    return out, {k: v[1].decode("utf-8", "replace") for k, v in attrs.items() if v[0] == "string"}


def read_pfm(path):
    """→ rgb float32 [H, W, 3], top row first (PF colour or Pf grey; the sign of the scale line gives the byte order)."""
    buf = open(path, "rb").read()
    parts, off = [], 0
    while len(parts) < 4:  # magic, width, height, scale — whitespace separated
        while buf[off:off + 1].isspace():
            off += 1
        e = off
        while not buf[e:e + 1].isspace():
            e += 1
        parts.append(buf[off:e].decode()); off = e
    off += 1
    magic, W, H, scale = parts[0], int(parts[1]), int(parts[2]), float(parts[3])
    if magic not in ("PF", "Pf"):
        raise ValueError("%s: not a PFM file" % path)
    c = 3 if magic == "PF" else 1
    img = np.frombuffer(buf, "<f4" if scale < 0 else ">f4", W * H * c, off).reshape(H, W, c)[::-1].astype(np.float32)
    return np.repeat(img, 3, 2) if c == 1 else np.ascontiguousarray(img)


def read_hdr(path):
    """Radiance RGBE (.hdr / .pic), flat or new-style run-length encoded scanlines, -Y H +X W orientation → rgb float32 [H, W, 3]."""
    buf = open(path, "rb").read()
    if not (buf.startswith(b"#?RADIANCE") or buf.startswith(b"#?RGBE")):
        raise ValueError("%s: not a Radiance HDR file" % path)
    e = buf.index(b"\n\n") + 2
    nl = buf.index(b"\n", e)
    res = buf[e:nl].split()
    if len(res) != 4 or res[0] != b"-Y" or res[2] != b"+X":
        raise ValueError("%s: only the standard -Y H +X W orientation is supported" % path)
    H, W = int(res[1]), int(res[3])
    off = nl + 1
    rgbe = np.zeros((H, W, 4), np.uint8)
    for y in range(H):
        if 8 <= W < 32768 and buf[off] == 2 and buf[off + 1] == 2 and (buf[off + 2] << 8 | buf[off + 3]) == W:
            off += 4
            for ch in range(4):
                x = 0
                while x < W:
                    n = buf[off]; off += 1
                    if n > 128:
                        rgbe[y, x:x + n - 128, ch] = buf[off]; off += 1; x += n - 128
                    else:
                        rgbe[y, x:x + n, ch] = np.frombuffer(buf, np.uint8, n, off); off += n; x += n
        else:
            rgbe[y] = np.frombuffer(buf, np.uint8, 4 * W, off).reshape(W, 4); off += 4 * W
    ex = rgbe[..., 3].astype(np.int32)
    f = np.where(ex > 0, np.ldexp(np.float32(1.0), ex - (128 + 8)), np.float32(0)).astype(np.float32)
    return (rgbe[..., :3].astype(np.float32) * f[..., None]).astype(np.float32)


def read_image(path):
    """HDR image by extension: .exr, .pfm, .hdr / .pic → rgb float32 [H, W, 3]."""
    ext = path.lower().rsplit(".", 1)[-1]
    if ext == "exr":
        return read_exr(path)[0]
    if ext == "pfm":
        return read_pfm(path)
    if ext in ("hdr", "pic", "rgbe"):
        return read_hdr(path)
    raise ValueError("%s: unsupported image format (exr, pfm, hdr)" % path)


def write_pfm(path, rgb):
    img = np.ascontiguousarray(rgb, np.float32)
    H, W, _ = img.shape
    with open(path, "wb") as f:
        f.write(("PF\n%d %d\n-1.0\n" % (W, H)).encode())
        f.write(img[::-1].astype("<f4").tobytes())  # PFM stores the bottom row first
