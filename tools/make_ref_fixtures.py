#!/usr/bin/env python3
"""Dev-time tool: mine the reference's shipped renders for known answers (DATA, not source).

Reads  /root/reference/scenes/*/*.exr   (only available in the development container)
Writes tests/golden/ref_logs.json       parsed per-iteration log lines embedded in the EXR headers
                                        (written by mitsuba/src/films/hdrfilm.cpp:527-534; the lines
                                        themselves are printed by guided_path.cpp:1176-1186, 1325, 1376)
       tests/golden/ref_cbox_images.npz 64x64 block-averaged RGB of cbox.exr / cbox-improved.exr + mean RGB
       tests/golden/ref_kitchen_reference.npz  the pixels of kitchen-reference.exr (the reference's converged 700x400 KITCHEN render,
                                        half precision) + RMSE / MAPE of the reference's own kitchen.exr and kitchen-improved.exr
                                        against it: the "equal error" target of bench.py's time_to_rmse block

These are the only outputs of the reference integrator that exist for this path (SURVEY.md §4, §6,
§8(c)); the oracle is pinned against them statistically (tests/test_oracle_reference_pins.py).
"""
import json, os, re, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from exr_min import read_exr, read_header, attr_string  # noqa: E402

REF = "/root/reference/scenes"
OUT = os.path.join(HERE, "..", "tests", "golden")
SCENES = ["cbox/cbox", "cbox/cbox-improved", "kitchen/kitchen", "kitchen/kitchen-improved",
          "spaceship/spaceship", "spaceship/spaceship-improved"]

F = r"([-+0-9.eE#INFa-z]+)"


def fnum(s):
    s = s.strip().rstrip(".")
    if "#INF" in s or s in ("inf", "-inf"):
        return float("-inf") if s.startswith("-") else float("inf")
    return float(s)


def triple(s):
    a, b, c = [fnum(x) for x in s.split(",")]
    return [a, b, c]


def parse_log(log):
    res = {"iterations": []}
    m = re.search(r"Starting render job \((\d+)x(\d+), (\d+) cores", log)
    res["width"], res["height"], res["cores"] = int(m.group(1)), int(m.group(2)), int(m.group(3))
    cur = None
    for line in log.split("\n"):
        m = re.search(r"ITERATION (\d+), (\d+) passes", line)
        if m:
            cur = {"iter": int(m.group(1)), "passes": int(m.group(2))}
            res["iterations"].append(cur)
            continue
        m = re.search(r"FINAL (\d+) passes", line)
        if m and cur is not None:
            cur["final_passes"] = int(m.group(1))
        m = re.search(F + r" seconds, Total passes: (\d+), Var: " + F + ",", line)
        if m and cur is not None:
            cur.setdefault("seconds", []).append(fnum(m.group(1)))
            cur.setdefault("total_passes", []).append(int(m.group(2)))
            cur.setdefault("var", []).append(fnum(m.group(3)))
        for key, pat in (("depth", "Depth"), ("mean_radiance", "Mean radiance"),
                         ("node_count", "Node count"), ("stat_weight", r"Stat\. weight")):
            m = re.search(pat + r"\s*= \[(.*)\]", line)
            if m and cur is not None:
                cur[key] = triple(m.group(1))
        m = re.search(r"Render time: " + F + "([sm])", line)
        if m:
            res["render_time_s"] = fnum(m.group(1)) * (60.0 if m.group(2) == "m" else 1.0)
        m = re.search(r"Normal rays traced : " + F + " ([MG])", line)
        if m:
            res["rays"] = fnum(m.group(1)) * (1e9 if m.group(2) == "G" else 1e6)
        m = re.search(r"Average path length : " + F + r" \(" + F + " ([MG]) / " + F + " ([MG])", line)
        if m:
            res["avg_path_length"] = fnum(m.group(1))
            res["samples"] = fnum(m.group(4)) * (1e9 if m.group(5) == "G" else 1e6)
    return res


def main():
    logs = {}
    for s in SCENES:
        buf = open(os.path.join(REF, s + ".exr"), "rb").read()
        attrs, _ = read_header(buf)
        logs[os.path.basename(s)] = parse_log(attr_string(attrs, "log"))
    with open(os.path.join(OUT, "ref_logs.json"), "w") as f:
        json.dump({"source": "log attribute of /root/reference/scenes/*/*.exr (tag 2024_10_08)",
                   "generator": "tools/make_ref_fixtures.py", "scenes": logs}, f, indent=1, allow_nan=True)
    imgs = {}
    for s in ("cbox/cbox", "cbox/cbox-improved"):
        _, ch = read_exr(os.path.join(REF, s + ".exr"))
        rgb = np.stack([ch["R"], ch["G"], ch["B"]], -1).astype(np.float64)
        H, W, _ = rgb.shape
        b = 8
        small = rgb.reshape(H // b, b, W // b, b, 3).mean((1, 3))
        key = os.path.basename(s).replace("-", "_")
        imgs[key + "_block8"] = small.astype(np.float32)
        imgs[key + "_mean_rgb"] = rgb.mean((0, 1)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ref_cbox_images.npz"), **imgs)
    print({k: (v.shape, v.mean()) for k, v in imgs.items()})
    rgb = {}
    for n in ("kitchen-reference", "kitchen", "kitchen-improved"):
        _, ch = read_exr(os.path.join(REF, "kitchen", n + ".exr"))
        rgb[n] = np.stack([ch["R"], ch["G"], ch["B"]], -1).astype(np.float64)
    ref = rgb["kitchen-reference"]
    err = {}
    for n in ("kitchen", "kitchen-improved"):
        d = rgb[n] - ref
        err[n.replace("-", "_") + "_rmse"] = np.float64(np.sqrt((d * d).mean()))
        err[n.replace("-", "_") + "_mape"] = np.float64((np.abs(d) / (ref + 0.01)).mean())
    # the same errors over the pixels where this build's picture can be compared at all: six of the scene's 289 meshes are missing from
    # the reference checkout; tools/make_kitchen_mask.py located their footprint (20x20-pixel blocks)
    mpath = os.path.join(OUT, "kitchen_missing_mesh_mask.npz")
    if os.path.exists(mpath):
        mk = np.load(mpath)
        keep = ~np.kron(mk["mask_blocks"], np.ones((int(mk["block"]), int(mk["block"]), ), np.uint8)).astype(bool)
        err["mask_blocks"], err["mask_block"] = mk["mask_blocks"], mk["block"]
        for n in ("kitchen", "kitchen-improved"):
            d = (rgb[n] - ref)[keep]
            err[n.replace("-", "_") + "_rmse_unmasked"] = np.float64(np.sqrt((d * d).mean()))
            err[n.replace("-", "_") + "_mape_unmasked"] = np.float64((np.abs(d) / (ref[keep] + 0.01)).mean())
    # 50x50-pixel block means of the reference's two guided renders (8 x 14 blocks): what the GPU tests compare this build's pictures of the
    # same configurations with, block by block (tests/test_real_scenes.py)
    for n in ("kitchen", "kitchen-improved"):
        err[n.replace("-", "_") + "_blocks50"] = rgb[n].reshape(8, 50, 14, 50, 3).mean((1, 3)).astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "ref_kitchen_reference.npz"), rgb=ref.astype(np.float16), mean_rgb=ref.mean((0, 1)), **err)
    print("kitchen-reference", ref.shape, ref.mean((0, 1)), float(ref.max()), {k: v for k, v in err.items() if np.ndim(v) == 0})
    for k, v in logs.items():
        print(k, [(i["iter"], i["passes"]) for i in v["iterations"]], v.get("avg_path_length"))


if __name__ == "__main__":
    main()
