"""Host-side mirror of the reference's integrator interface for the GuidedPathTracer hot path.

`bindings`    ctypes view of the C-ABI in include/ppg.h (libppg_hip.so; the same classes can be pointed
              at the CPU oracle's ppgo_* symbols by tests/bench — never by product code)
`scenes`      procedural scene descriptions (CBOX = scenes/cbox/cbox.xml of the reference restated)
`mitsuba_xml` loader for the Mitsuba scene-XML subset of the bundled scenes (+ OBJ reader); `spectrum`: colour values
`integrator`  GuidedPathTracer: property names / defaults / render() semantics of guided_path.cpp
`distributed` tile-sharded multi-GPU driver (one process per GPU, RCCL all-reduce of SD-tree statistics)
"""
from .bindings import Engine, PPGError, Config, PassStats, TreeStats, hip_library_path  # noqa: F401
from .scenes import SceneDesc, cbox_scene, perspective_camera, perspective_camera_from_matrix, resize_camera, room_scene, torus_scene, save_scene, load_scene_file  # noqa: F401
from .mitsuba_xml import load_scene, load_obj, save_scene_xml, SceneError  # noqa: F401
from .integrator import GuidedPathTracer  # noqa: F401,E402
