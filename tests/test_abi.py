"""The C-ABI shared library loads without a GPU and exports every symbol include/ppg.h declares;
property parsing rejects what the reference asserts on (guided_path.cpp:1023-1080)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppg_[a-z_0-9]+)\s*\(", src)))


def test_hip_library_exports_every_declared_symbol(hip_lib_path):
    lib = C.CDLL(hip_lib_path)
    names = _declared("ppg.h") + _declared("ppg_testhooks.h")  # (the boundary; the tests' own hook: kept out of the boundary's header)
    assert len(names) >= 30 and "ppg_debug_build_bvh" not in _declared("ppg.h")
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    lib.ppg_description.restype = C.c_char_p
    assert lib.ppg_description() == b"Guided path tracer"  # MTS_EXPORT_PLUGIN(GuidedPathTracer, "Guided path tracer"), GP:2422


def test_config_defaults_are_the_references(hip_lib_path):
    from ppg_host.bindings import Config
    lib = C.CDLL(hip_lib_path)
    cfg = Config()
    lib.ppg_config_default(C.byref(cfg))
    expect = dict(nee=b"never", sampleCombination=b"automatic", spatialFilter=b"nearest", directionalFilter=b"nearest",
                  bsdfSamplingFractionLoss=b"none", sdTreeMaxMemory=-1, sTreeThreshold=12000, sppPerPass=4, budgetType=b"seconds",
                  dumpSDTree=0, rrDepth=5, maxDepth=-1, strictNormals=0, hideEmitters=0)  # GP:1015-1084, integrator.cpp:192-218
    for k, v in expect.items():
        assert getattr(cfg, k) == v, k
    assert abs(cfg.dTreeThreshold - 0.01) < 1e-9 and cfg.bsdfSamplingFraction == 0.5 and cfg.budget == 300.0
    for k, v in Config.DEFAULTS.items():  # the Python mirror carries the same defaults
        got = getattr(cfg, k)
        got = got.decode() if isinstance(got, bytes) else got
        if isinstance(v, float):
            assert abs(got - v) < 1e-6, k
        else:
            assert got == v, k


@pytest.mark.parametrize("field,value", [("nee", "sometimes"), ("sampleCombination", "average"), ("spatialFilter", "gauss"),
                                         ("directionalFilter", "stochastic"), ("bsdfSamplingFractionLoss", "l2"), ("budgetType", "minutes")])
def test_unknown_enum_strings_are_rejected(oracle_lib, field, value):
    import ppg_host
    with pytest.raises(ppg_host.PPGError) as ei:  # the reference: Assert(false) in the constructor
        ppg_host.Engine(oracle_lib, "ppgo_", **{field: value})
    assert ei.value.code == -1 and field in str(ei.value)


def test_missing_library_fails_loudly(tmp_path):
    import ppg_host
    with pytest.raises(FileNotFoundError):
        ppg_host.Engine(str(tmp_path / "libppg_hip.so"))


def test_unknown_property_name_is_an_error():
    from ppg_host.bindings import Config
    with pytest.raises(KeyError):
        Config.make(sppPerPas=4)


def test_oracle_rejects_unknown_bsdf_type(oracle_lib):
    import ppg_host
    scene = ppg_host.cbox_scene(8, 8)
    scene.materials[0]["type"] = 99
    e = ppg_host.Engine(oracle_lib, "ppgo_", budgetType="spp", budget=4)
    with pytest.raises(ppg_host.PPGError):
        e.set_scene(scene)


def test_ctypes_mirrors_have_the_layout_of_the_header(tmp_path):
    """include/ppg.h compiled by gcc vs. ppg_host.bindings: sizes of the structures crossing the boundary and the offsets of the fields
    appended over time (a drop-in boundary must not drift between its two descriptions)."""
    import subprocess
    from ppg_host import bindings as b
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ppg.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", '
                   'sizeof(ppg_material), sizeof(ppg_sphere), sizeof(ppg_emitter), sizeof(ppg_camera), sizeof(ppg_scene), offsetof(ppg_material, rtrans), '
                   'offsetof(ppg_scene, environment), offsetof(ppg_scene, rtrans), offsetof(ppg_scene, spheres), offsetof(ppg_sphere, material), sizeof(ppg_envmap), offsetof(ppg_envmap, scale), offsetof(ppg_scene, envmap)); return 0; }\n')
    exe = str(tmp_path / "layout")
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", exe, str(src)], check=True)
    got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(b.Material), C.sizeof(b.Sphere), C.sizeof(b.Emitter), C.sizeof(b.Camera), C.sizeof(b.Scene), b.Material.rtrans.offset,
            b.Scene.environment.offset, b.Scene.rtrans.offset, b.Scene.spheres.offset, b.Sphere.material.offset, C.sizeof(b.EnvMap), b.EnvMap.scale.offset,
            b.Scene.envmap.offset]
    assert got == want and got[:2] == [80, 64]
