"""The C ABI driven by a host other than the one in this repository: a scene description put
together with ctypes.  Covers `PatternType::Blend` (src/pattern.rs:177-189), which the reference's
JSON loader cannot produce (src/scene.rs:350) and therefore no scene file reaches."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu


class RtLight(C.Structure):
    _fields_ = [("intensity", C.c_double * 3), ("position", C.c_double * 3)]


class RtMaterial(C.Structure):
    _fields_ = [("color", C.c_double * 3), ("ambient", C.c_double), ("diffuse", C.c_double), ("specular", C.c_double),
                ("shininess", C.c_double), ("reflective", C.c_double), ("refractive_index", C.c_double),
                ("transparency", C.c_double), ("pattern", C.c_int32), ("reserved", C.c_int32)]


class RtPattern(C.Structure):
    _fields_ = [("type", C.c_int32), ("first", C.c_int32), ("second", C.c_int32), ("reserved", C.c_int32),
                ("inv", C.c_double * 16), ("a", C.c_double * 3), ("b", C.c_double * 3)]


class RtSceneDesc(C.Structure):
    _fields_ = [("light", RtLight), ("n_leaves", C.c_int32), ("n_gates", C.c_int32), ("n_xforms", C.c_int32),
                ("n_triangles", C.c_int32), ("n_materials", C.c_int32), ("n_patterns", C.c_int32),
                ("n_csg", C.c_int32), ("n_chain_mats", C.c_int32), ("csg", C.c_void_p), ("leaves", C.c_void_p),
                ("gates", C.c_void_p), ("xforms", C.c_void_p), ("triangles", C.c_void_p),
                ("materials", C.POINTER(RtMaterial)), ("patterns", C.POINTER(RtPattern)),
                ("chain_mats", C.c_void_p), ("xform_chain", C.c_void_p), ("gate_chain", C.c_void_p)]


RT_PAT_BLEND = 7


def scene_text(root, pattern):
    base = json.load(open(os.path.join(root, "scenes", "example.json")))
    scale = [0.4, 0, 0, 0, 0, 0.4, 0, 0, 0, 0, 0.4, 0, 0, 0, 0, 1]
    shapes = [{"ty": "sphere", "material": {"color": [1, 1, 1], "pattern": dict(pattern, transform=scale)}},
              {"ty": "plane", "transform": [1, 0, 0, 0, 0, 1, 0, -1, 0, 0, 1, 0, 0, 0, 0, 1],
               "material": {"color": [0.5, 0.6, 0.7]}}]
    return json.dumps(dict(base, shapes=shapes, canvas_width=96, canvas_height=54))


def test_blend_pattern_through_a_hand_built_description(pkg, root):
    amd, host = pkg.libraries()
    stripe = {"ty": "stripe", "primary_color": [1.0, 0.1, 0.1], "secondary_color": [0.1, 0.1, 1.0]}
    ring = {"ty": "ring", "primary_color": [0.2, 0.9, 0.2], "secondary_color": [0.9, 0.9, 0.1]}
    a = pkg.Scene(scene_text(root, stripe), root, device=None)  # host side only: parsing and flattening
    b = pkg.Scene(scene_text(root, ring), root, device=None)
    host.rth_scene_desc.restype = C.POINTER(RtSceneDesc)
    host.rth_scene_desc.argtypes = [C.c_void_p]
    da = host.rth_scene_desc(a._h).contents
    assert da.n_patterns == 1 and da.n_materials == 2
    pat_a = RtPattern.from_buffer_copy(da.patterns[0])
    mats = (RtMaterial * da.n_materials)(*[RtMaterial.from_buffer_copy(da.materials[i]) for i in range(da.n_materials)])
    pat_b = RtPattern.from_buffer_copy(host.rth_scene_desc(b._h).contents.patterns[0])
    da = host.rth_scene_desc(a._h).contents  # the pointer is per thread: fetch A's arrays again
    blend = RtPattern()
    blend.type, blend.first, blend.second = RT_PAT_BLEND, 0, 1
    for i in range(16):
        blend.inv[i] = pat_a.inv[i]  # the blend's own transform; the operands' transforms play no part (pattern.rs:177-189)
    patterns = (RtPattern * 3)(pat_a, pat_b, blend)
    for i in range(len(mats)):
        if mats[i].pattern == 0:
            mats[i].pattern = 2
    desc = RtSceneDesc.from_buffer_copy(da)
    desc.n_patterns, desc.patterns = 3, C.cast(patterns, C.POINTER(RtPattern))
    desc.materials = C.cast(mats, C.POINTER(RtMaterial))

    amd.rt_scene_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    scene = C.c_void_p()
    assert amd.rt_scene_create(C.byref(desc), 0, C.byref(scene)) == 0, amd.rt_last_error()
    cam = a.camera()
    w, h = cam.hsize, cam.vsize
    rgb64 = np.zeros((h, w, 3), np.float64)
    leaf = np.zeros((h, w), np.int32)
    st = pkg.RtStats()
    rc = amd.rt_render_host(scene, C.byref(cam), 0, 1, 0, None, rgb64.ctypes.data_as(C.c_void_p),
                            leaf.ctypes.data_as(C.c_void_p), C.byref(st))
    assert rc == 0, amd.rt_last_error()
    amd.rt_scene_destroy.argtypes = [C.c_void_p]
    amd.rt_scene_destroy(scene)

    # lighting is linear in the pattern colour, so the blend renders to the mean of the two operand renders
    ra = orc.OracleScene(scene_text(root, stripe), root).render(jobs=4)
    rb = orc.OracleScene(scene_text(root, ring), root).render(jobs=4)
    assert np.array_equal(leaf, ra["hit_leaf"])
    assert np.abs(rgb64 - 0.5 * (ra["rgb64"] + rb["rgb64"])).max() <= 1e-12
    on_sphere = leaf == 0
    assert on_sphere.sum() > 100 and np.abs(ra["rgb64"] - rb["rgb64"])[on_sphere].max() > 0.1  # the operands do differ

    bad = RtPattern.from_buffer_copy(blend)
    bad.second = 7
    patterns[2] = bad
    assert amd.rt_scene_create(C.byref(desc), 0, C.byref(scene)) == -1 and b"blend" in amd.rt_last_error()


def test_plain_c_program_renders_the_example_scene(pkg, root, tmp_path):
    """tests/abi_smoke.c (C99, built by the Makefile): create -> render -> destroy on example.json's two
    shapes flattened by hand; its frame must equal the one the host library's loader and flattener give."""
    import subprocess
    scene = pkg.Scene.from_file(os.path.join(root, "scenes", "example.json"), base_dir=root, device=0)
    scene.set_canvas(96, 54)
    want = scene.render(rgb64=False, hit_leaf=False)[0]["rgb8"]
    cam = scene.camera()
    (tmp_path / "camera.bin").write_bytes(bytes(cam))
    exe = os.path.join(root, "ray-tracer_amd", "lib", "abi_smoke")
    p = subprocess.run([exe, str(tmp_path / "camera.bin"), str(tmp_path / "out.rgb8")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    got = np.frombuffer((tmp_path / "out.rgb8").read_bytes(), np.uint8).reshape(54, 96, 3)
    assert np.array_equal(got, want)
