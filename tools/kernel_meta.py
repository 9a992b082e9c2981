#!/usr/bin/env python3
"""Code-object metadata of the kernels in a built library or object file: VGPRs, SGPRs, spills, scratch, LDS, code size.

    python tools/kernel_meta.py [practical-path-guiding_amd/lib/libppg_hip.so | build/*.o ...] [--filter k_shade]

Needs no GPU: it unbundles the gfx950 code objects (clang-offload-bundler) and reads their notes (llvm-readelf --notes) and symbol
sizes.  Used to check register budgets of kernel variants before spending GPU time on them."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(path, tmp):
    """gfx950 code objects bundled in `path` (a .so / .o produced by hipcc)"""
    out = []
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = 0
    k = 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            break
        n = int.from_bytes(data[i + 24:i + 32], "little")
        off = i + 32
        for _ in range(n):
            eo = int.from_bytes(data[off:off + 8], "little")
            es = int.from_bytes(data[off + 8:off + 16], "little")
            ts = int.from_bytes(data[off + 16:off + 24], "little")
            triple = data[off + 24:off + 24 + ts].decode()
            off += 24 + ts
            if "gfx" in triple and es:
                f = os.path.join(tmp, "co%d.elf" % k)
                k += 1
                open(f, "wb").write(data[i + eo:i + eo + es])
                out.append(f)
        pos = i + 1
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.split("\n")


def kernels(elf):
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf], capture_output=True, text=True).stdout
    syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", elf], capture_output=True, text=True).stdout
    sizes = {}
    for line in syms.splitlines():
        f = line.split()
        if len(f) >= 8 and f[3] == "FUNC":
            sizes[f[7]] = int(f[2])
    res = []
    cur = {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip()
        if key == "agpr_count" or (key == "args" and cur.get("name")):
            pass
        if key in ("name", "symbol", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                   "group_segment_fixed_size", "agpr_count"):
            cur[key] = val.strip("'\"")
        if key == "wavefront_size":
            if "symbol" in cur:
                res.append(cur)
            cur = {}
    for k in res:
        k["code_bytes"] = sizes.get(k.get("name", ""), 0)
    return res


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    flt = None
    if "--filter" in sys.argv:
        flt = sys.argv[sys.argv.index("--filter") + 1]
        args = [a for a in args if a != flt]
    if not args:
        args = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "practical-path-guiding_amd", "lib", "libppg_hip.so")]
    with tempfile.TemporaryDirectory() as tmp:
        rows = []
        for a in args:
            for co in code_objects(a, tmp):
                rows += kernels(co)
        names = demangle([r.get("name", "?") for r in rows])
        print("%-72s %5s %5s %6s %6s %8s %7s %8s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "scratch", "lds", "code"))
        seen = set()
        for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
            n = re.sub(r"^void ", "", n)
            n = re.sub(r"\(.*$", "", n)
            if flt and flt not in n:
                continue
            line = "%-72s %5s %5s %6s %6s %8s %7s %8s" % (n[:72], r.get("vgpr_count"), r.get("sgpr_count"), r.get("vgpr_spill_count"), r.get("sgpr_spill_count"),
                                                        r.get("private_segment_fixed_size"), r.get("group_segment_fixed_size"), r.get("code_bytes"))
            if line not in seen:  # static kernels are compiled into every translation unit
                seen.add(line)
                print(line)


if __name__ == "__main__":
    main()
