"""`python -m ppg_host scene.xml [-o out.exr] [-D name=value ...] [-P property=value ...]`

The role of `mitsuba scene.xml -o out.exr -D ...` (mitsuba.cpp:58-87, 300-400) for this path: load the scene-XML
subset (mitsuba_xml.py), render with the guided path tracer on the GPU, print the reference's per-iteration log
lines (GP:1176-1186, 1321-1326, 1376), write an EXR (render log attached like hdrfilm.cpp:525-533) or a PFM.
`--ppgs FILE` only converts the scene to the flat binary that the C++ driver `bin/ppg_render` reads.
"""
import argparse
import os
import sys
import time


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m ppg_host", description=__doc__.split("\n\n")[0])
    ap.add_argument("scene", help="Mitsuba scene XML (guided_path integrator, supported subset)")
    ap.add_argument("-o", "--output", help="output image (.exr or .pfm); default: scene name + .exr")
    ap.add_argument("-D", dest="defines", action="append", default=[], metavar="name=value", help="define $name for the XML (like mitsuba -D)")
    ap.add_argument("-P", dest="props", action="append", default=[], metavar="property=value", help="override an integrator property")
    ap.add_argument("--size", metavar="WxH", help="override the film size")
    ap.add_argument("--ppgs", metavar="FILE", help="write the flattened scene for bin/ppg_render and exit")
    ap.add_argument("--lenient", action="store_true", help="replace unsupported BSDFs by diffuse(0.5) instead of failing")
    ap.add_argument("--data-dir", help="`data` directory of a Mitsuba tree (roughplastic reads data/microfacet/*.dat); default $PPG_MITSUBA_DATA")
    ap.add_argument("--constant-env", metavar="R,G,B", help="add a constant environment emitter (stand-in lighting when --lenient skipped the scene's own, e.g. sunsky)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("-q", "--quiet", action="store_true")
    a = ap.parse_args(argv)

    import torch  # noqa: F401  (its bundled HIP runtime must be loaded before libppg_hip.so, see tests/conftest.py)
    from . import GuidedPathTracer, load_scene, save_scene
    from .mitsuba_xml import GUIDED_PATH_PROPS
    from .imageio import write_exr, write_pfm

    defines = dict(d.split("=", 1) for d in a.defines)
    w = h = None
    if a.size:
        w, h = (int(v) for v in a.size.lower().split("x"))
    desc, props, info = load_scene(a.scene, defines, strict=not a.lenient, width=w, height=h, data_dir=a.data_dir)
    if a.constant_env:
        if desc.environment is not None or desc.envmap is not None:
            ap.error("--constant-env: the scene already has an environment emitter")
        desc.environment = tuple(float(v) for v in a.constant_env.split(","))
    for kv in a.props:
        k, v = kv.split("=", 1)
        if k not in GUIDED_PATH_PROPS:
            ap.error("unknown integrator property %r" % k)
        t = GUIDED_PATH_PROPS[k]
        props[k] = (v.lower() in ("1", "true")) if t is int and v.lower() in ("true", "false", "0", "1") else t(v)
    say = (lambda *s: None) if a.quiet else (lambda *s: print(*s, file=sys.stderr, flush=True))
    for wmsg in info["warnings"]:
        say("warning:", wmsg)
    say("scene: %d triangles, %d materials, %d emitters, film %dx%d" % (desc.n_triangles, len(desc.materials), len(desc.emitters), info["width"], info["height"]))
    if a.ppgs:
        save_scene(desc, a.ppgs)
        with open(a.ppgs + ".props", "w") as f:
            for k, v in props.items():
                f.write("%s=%s\n" % (k, v))
        say("wrote", a.ppgs, "and", a.ppgs + ".props")
        return 0

    log_lines = []

    def log(rec):  # the reference's log lines, GP:1325-1326 and GP:1176-1186
        t = rec["tree"]
        for st in rec["stats"]:
            ttuv = st["seconds"] * st["variance"]
            stuv = st["passes_rendered_local"] * props.get("sppPerPass", 4) * st["variance"]
            line = "%.2f seconds, Total passes: %d, Var: %f, TTUV: %f, STUV: %f." % (st["seconds"], st["passes_rendered_total"], st["variance"], ttuv, stuv)
            log_lines.append(line); say(line)
        line = ("Distribution statistics:\n  Depth         = [%d, %f, %d]\n  Mean radiance = [%f, %f, %f]\n  Node count    = [%d, %f, %d]\n"
                "  Stat. weight  = [%f, %f, %f]\n" % (t["min_depth"], t["avg_depth"], t["max_depth"], t["min_mean_radiance"], t["avg_mean_radiance"],
                                                     t["max_mean_radiance"], t["min_nodes"], t["avg_nodes"], t["max_nodes"], t["min_stat_weight"],
                                                     t["avg_stat_weight"], t["max_stat_weight"]))
        log_lines.append(line); say(line)

    out = a.output or os.path.splitext(a.scene)[0] + ".exr"
    if props.get("dumpSDTree") and not props.get("dumpPrefix"):
        props["dumpPrefix"] = os.path.splitext(out)[0]  # "<dest>-%02d.sdt", GP:1192-1195
    gpt = GuidedPathTracer(log=log, device=a.device, **props)
    t0 = time.monotonic()
    img = gpt.render(desc)
    dt = time.monotonic() - t0
    spp = sum(s["samples"] for it in gpt.iterations for s in it["stats"]) / float(info["width"] * info["height"])
    line = "Render time: %.3f s (%.1f spp, %.2f Msamples/s)" % (dt, spp, spp * info["width"] * info["height"] / dt / 1e6)
    log_lines.append(line); say(line)
    if out.lower().endswith(".pfm"):
        write_pfm(out, img)
    else:
        if not out.lower().endswith(".exr"):
            out += ".exr"
        write_exr(out, img, {"log": "\n".join(log_lines), "generatedBy": "practical-path-guiding_amd guided_path (MI355X)"})
    say("Writing image to \"%s\" .." % out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
