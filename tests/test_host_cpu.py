"""CPU tests of the host side and of the C ABI surface (no GPU, no compute calls)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(pkg):
    amd, host = pkg.libraries()
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    names = set(re.findall(r"\b(rt_[a-z_]+)\s*\(", header))
    assert {"rt_scene_create", "rt_render", "rt_color_at", "rt_last_error"} <= names
    for n in sorted(names):
        assert hasattr(amd, n), "librt_amd.so does not export %s" % n
    for n in ("rth_scene_from_json", "rth_parallel_render", "rth_ppm_encode", "rth_scene_camera"):
        assert hasattr(host, n)


def test_rust_binding_declares_every_export():
    """rust/src/ffi.rs (not compiled here: no Rust toolchain) names every function the header declares."""
    header = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    ffi = open(os.path.join(ROOT, "rust", "src", "ffi.rs")).read()
    names = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", header))
    declared = set(re.findall(r"pub fn (rt_[a-z_0-9]+)\(", ffi))
    assert names - declared == set(), "missing in ffi.rs: %s" % sorted(names - declared)
    assert declared - names == set(), "ffi.rs declares what the header does not: %s" % sorted(declared - names)


def test_header_is_plain_c99(tmp_path):
    """include/rt_amd.h compiles as C99 with no extensions; the smoke program that uses it was built the same way."""
    import subprocess
    src = tmp_path / "use_header.c"
    src.write_text('#include "rt_amd.h"\nint main(void) { rt_scene_desc d; rt_camera c; (void)d; (void)c; return (int)sizeof(rt_stats) - 128; }\n')
    exe = tmp_path / "use_header"
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    assert subprocess.run([str(exe)]).returncode == 0
    assert os.path.exists(os.path.join(ROOT, "ray-tracer_amd", "lib", "abi_smoke"))


def test_struct_layouts_match_the_header(pkg):
    assert ctypes.sizeof(pkg.RtCamera) == 8 + 3 * 8 + 16 * 8
    assert ctypes.sizeof(pkg.RtStats) == 4 * 8 + 2 * 8 + 2 * 8 + 8 * 8


@pytest.mark.parametrize("name", ["example", "egg", "glass", "pattern", "teapot"])
def test_flattening_agrees_with_the_oracle_tree(pkg, name):
    path = os.path.join(ROOT, "scenes", name + ".json")
    s = pkg.Scene.from_file(path, base_dir=ROOT, device=None)
    o = orc.OracleScene.from_file(path, base_dir=ROOT)
    info = s.info()
    assert (info["width"], info["height"]) == (o.info()["width"], o.info()["height"])
    assert info["leaves"] == o.info()["leaves"]  # same depth-first numbering of non-container shapes


def test_camera_description(pkg):
    s = pkg.Scene.from_file(os.path.join(ROOT, "scenes", "example.json"), base_dir=ROOT, device=None)
    s.set_canvas(201, 101)
    cam = s.camera()
    half_view = np.tan(1.0472 / 2.0)
    assert cam.hsize == 201 and cam.vsize == 101
    assert cam.half_width == half_view and cam.half_height == half_view / (201 / 101)
    assert cam.pixel_size == half_view * 2.0 / 201
    # the camera sits at camera_from: last column of the inverse view transform
    assert [cam.inv[3], cam.inv[7], cam.inv[11]] == pytest.approx([0.0, 1.5, -10.0], abs=1e-12)
    assert list(cam.inv[12:16]) == pytest.approx([0, 0, 0, 1], abs=1e-12)


def test_ppm_bytes_match_canvas_save(pkg):
    rng = np.random.default_rng(5)
    rgb = rng.uniform(-0.2, 1.3, (7, 23, 3))
    rgb[0, 0] = [0.0, 1.0, 1e-9]
    want = orc.ppm_encode(rgb)
    got = pkg.ppm_encode(orc.quantise(rgb))
    assert got == want
    assert got.startswith(b"P3\n23 7\n255\n") and got.endswith(b"\n")


def test_loader_errors_keep_the_reference_messages(pkg):
    base = json.load(open(os.path.join(ROOT, "scenes", "example.json")))

    def load(**kw):
        return pkg.Scene(json.dumps(dict(base, **kw)), ROOT, device=None)

    with pytest.raises(pkg.RtError, match="Unrecognized shape type"):
        load(shapes=[{"ty": "torus"}])
    with pytest.raises(pkg.RtError, match="Shape transform should have an inverse"):
        load(shapes=[{"ty": "sphere", "transform": [1, 0, 0, 0]}])
    with pytest.raises(pkg.RtError, match="Unrecognized pattern"):
        load(shapes=[{"ty": "sphere", "material": {"color": [1, 1, 1], "pattern": {"ty": "blend"}}}])
    with pytest.raises(pkg.RtError, match="Model requires a path"):
        load(shapes=[{"ty": "model"}])
    with pytest.raises(pkg.RtError, match="Failed to open file"):
        load(shapes=[{"ty": "model", "path": "models/absent.obj"}])
    with pytest.raises(pkg.RtError, match="missing field"):
        pkg.Scene('{"canvas_width": 10}', ROOT, device=None)
    with pytest.raises(pkg.RtError, match="CSG union must have at least two operands"):
        load(shapes=[{"ty": "difference", "children": [{"ty": "sphere"}]}])


def test_csg_operators_flatten(pkg):
    """CSG operators are transform levels without a gate; their leaves stay in depth-first order
    and agree with the oracle's tree."""
    path = os.path.join(ROOT, "tests", "golden", "csg.json")
    s = pkg.Scene.from_file(path, base_dir=ROOT, device=None)
    ref = orc.OracleScene.from_file(path, base_dir=ROOT)
    assert s.info()["leaves"] == ref.info()["leaves"] == 16
    assert s.info()["gates"] == 2  # the bowl group and the group below its difference operator


def test_empty_groups_and_empty_shapes_flatten(pkg):
    base = json.load(open(os.path.join(ROOT, "scenes", "example.json")))
    shapes = [{"ty": "group"}, {"ty": "group", "children": [{"ty": "empty"}, {"ty": "group", "children": []}]},
              {"ty": "sphere"}]
    s = pkg.Scene(json.dumps(dict(base, shapes=shapes)), ROOT, device=None)
    info = s.info()
    assert info["leaves"] == 2 and info["gates"] == 1  # childless groups vanish (src/shape.rs:1186-1188)


def test_rendering_without_a_device_fails_loudly(pkg):
    s = pkg.Scene.from_file(os.path.join(ROOT, "scenes", "example.json"), base_dir=ROOT, device=None)
    with pytest.raises(pkg.RtError):
        s.render()


@pytest.mark.parametrize("n_bands,n_dirs", [(20, 100), (300, 20000), (3000, 6000)])
def test_cone_band_direction_index_is_conservative(pkg, n_bands, n_dirs):
    """Every cone whose parallel-ray branch a direction triggers must be listed in that direction's
    grid cell (host-only check of the index the kernels look up; half the directions lie on a cone)."""
    amd, _ = pkg.libraries()
    fn = amd.rt_selftest_band_grid
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]
    n, entries = ctypes.c_int(), ctypes.c_uint64()
    missing = fn(n_bands, n_dirs, 0xC0DE + n_bands, ctypes.byref(n), ctypes.byref(entries))
    assert missing == 0
    if n_bands <= 32:
        assert n.value == 0  # short lists are scanned, not indexed
    else:
        assert n.value >= 32 and entries.value > 0


@pytest.mark.parametrize("n_items", [0, 1, 5000, 100000])
def test_bvh_build_does_not_depend_on_the_thread_count(pkg, n_items):
    """Large ranges are built by several threads and spliced together in the order one thread would
    have produced: the structure (every node, the primitive orders, the stack depth) must come out
    byte for byte the same, so image parity never depends on the machine's core count."""
    amd, _ = pkg.libraries()
    fn = amd.rt_selftest_bvh_build
    fn.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64),
                   ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_double)]
    results = {}
    for threads in (1, 2, 5, 8):
        h, nodes, sec = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_double()
        assert fn(n_items, 0xB0B + n_items, threads, ctypes.byref(h), ctypes.byref(nodes), ctypes.byref(sec)) == 0
        results[threads] = (h.value, nodes.value)
        print("items", n_items, "threads", threads, "nodes", nodes.value, "seconds %.4f" % sec.value)
    assert len(set(results.values())) == 1
    if n_items >= 5000:
        assert results[1][1] > n_items // 16


def test_transform_chains_are_listed_only_where_the_product_is_inexact(pkg):
    """A ray reaches a nested shape through one transform_inverse per level (src/shape.rs:1467-1470).
    The flattener lists those matrices for the device when two or more differ from the identity;
    with fewer the single product is the same ray bit for bit and no chain is emitted."""
    class Ref(ctypes.Structure):
        _fields_ = [("first", ctypes.c_int32), ("count", ctypes.c_int32)]

    class Desc(ctypes.Structure):  # rt_scene_desc of include/rt_amd.h
        _fields_ = [("light", ctypes.c_double * 6)] + [(n, ctypes.c_int32) for n in (
            "n_leaves", "n_gates", "n_xforms", "n_triangles", "n_materials", "n_patterns", "n_csg", "n_chain_mats")] + [
            (n, ctypes.c_void_p) for n in ("csg", "leaves", "gates", "xforms", "triangles", "materials", "patterns")] + [
            ("chain_mats", ctypes.POINTER(ctypes.c_double)), ("xform_chain", ctypes.POINTER(Ref)),
            ("gate_chain", ctypes.POINTER(Ref))]

    _, host = pkg.libraries()
    host.rth_scene_desc.restype = ctypes.POINTER(Desc)
    host.rth_scene_desc.argtypes = [ctypes.c_void_p]
    base = json.load(open(os.path.join(ROOT, "scenes", "example.json")))
    move = [1, 0, 0, 2, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1]
    grow = [2, 0, 0, 0, 0, 3, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 1]
    turn = [0, 0, 1, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 0, 1]
    shapes = [
        {"ty": "sphere", "transform": move},                                                     # top level: none
        {"ty": "group", "transform": move, "children": [{"ty": "cube"}]},                        # one matrix: none
        {"ty": "group", "children": [{"ty": "cube", "transform": grow}]},                        # one matrix: none
        {"ty": "group", "transform": move, "children": [                                         # gate: none
            {"ty": "cube", "transform": grow},                                                   # two: chain
            {"ty": "group", "transform": turn, "children": [{"ty": "sphere", "transform": grow},  # gate two, leaf three
                                                            {"ty": "sphere"}]}]},                # two
        {"ty": "difference", "transform": turn, "children": [{"ty": "sphere", "transform": move}, {"ty": "cube"}]},
    ]
    s = pkg.Scene(json.dumps(dict(base, shapes=shapes)), ROOT, device=None)
    d = host.rth_scene_desc(s._h).contents
    assert d.n_xforms == 8 and d.n_gates == 4
    assert [d.xform_chain[i].count for i in range(8)] == [0, 0, 0, 2, 3, 2, 2, 0]
    assert [d.gate_chain[i].count for i in range(4)] == [0, 0, 0, 2]
    assert d.n_chain_mats == 2 + 3 + 2 + 2 + 2
    inv = lambda m: np.linalg.inv(np.array(m, float).reshape(4, 4))
    f = d.xform_chain[4].first  # the sphere three levels down: move, turn, grow, outermost first
    got = np.array([d.chain_mats[16 * f + k] for k in range(48)]).reshape(3, 4, 4)
    for g, want in zip(got, (inv(move), inv(turn), inv(grow))):
        assert np.allclose(g, want, atol=1e-15)
    # the shipped scenes, the teapot's transformed model with its identity OBJ groups included, need none
    for name in ("example", "egg", "glass", "pattern", "teapot"):
        t = pkg.Scene.from_file(os.path.join(ROOT, "scenes", name + ".json"), base_dir=ROOT, device=None)
        assert host.rth_scene_desc(t._h).contents.n_chain_mats == 0, name


@pytest.mark.parametrize("seed", range(25))
def test_random_scenes_flatten_like_the_oracle_tree(pkg, seed, tmp_path):
    """The host loader and the oracle's loader read the same random JSON (nested groups, CSG pairs,
    every primitive name, models from a generated OBJ file) into the same number of leaves."""
    import test_gpu_fuzz as fuzz
    for text in (fuzz.random_scene(seed), fuzz.mesh_scene(seed, str(tmp_path / "soup.obj"))):
        s = pkg.Scene(text, ROOT, device=None)
        ref = orc.OracleScene(text, base_dir=ROOT)
        assert s.info()["leaves"] == ref.info()["leaves"]
        assert (s.info()["width"], s.info()["height"]) == (ref.info()["width"], ref.info()["height"])


@pytest.mark.parametrize("seed", range(12))
def test_stored_world_normals_equal_normal_at(pkg, seed, tmp_path):
    """Planes and flat triangles have one normal wherever they are hit.  The library stores
    shape::normal_at's result for them (src/shape.rs:1478-1484: transposition(full_inv) * normal, w dropped,
    normalized) when a scene is created and the kernels load it instead of recomputing it per hit: the stored
    value must equal the oracle's normal_at bit for bit, under random transforms nested three groups deep."""
    import test_gpu_fuzz as fuzz

    class Leaf(ctypes.Structure):  # rt_leaf of include/rt_amd.h
        _fields_ = [(n, ctypes.c_int32) for n in ("type", "closed", "material", "xform", "gate", "tri", "csg", "csg_left")] + [
            ("minimum", ctypes.c_double), ("maximum", ctypes.c_double), ("plane_normal", ctypes.c_double * 3)]

    class Xform(ctypes.Structure):  # rt_xform
        _fields_ = [("ray_inv", ctypes.c_double * 16), ("full_inv", ctypes.c_double * 16), ("fwd", ctypes.c_double * 16)]

    class Triangle(ctypes.Structure):  # rt_triangle
        _fields_ = [(n, ctypes.c_double * 3) for n in ("p1", "e1", "e2", "normal", "n1", "n2", "n3")]

    class Desc(ctypes.Structure):  # rt_scene_desc
        _fields_ = [("light", ctypes.c_double * 6)] + [(n, ctypes.c_int32) for n in (
            "n_leaves", "n_gates", "n_xforms", "n_triangles", "n_materials", "n_patterns", "n_csg", "n_chain_mats")] + [
            ("csg", ctypes.c_void_p), ("leaves", ctypes.POINTER(Leaf)), ("gates", ctypes.c_void_p), ("xforms", ctypes.POINTER(Xform)),
            ("triangles", ctypes.POINTER(Triangle))]

    assert ctypes.sizeof(Leaf) == 72 and ctypes.sizeof(Xform) == 384 and ctypes.sizeof(Triangle) == 168
    amd, host = pkg.libraries()
    host.rth_scene_desc.restype = ctypes.POINTER(Desc)
    host.rth_scene_desc.argtypes = [ctypes.c_void_p]
    fn = amd.rt_selftest_world_normal
    fn.argtypes = [ctypes.POINTER(ctypes.c_double)] * 3
    olib = orc.lib()
    olib.orc_leaf_normal_at.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_double,
                                        ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(seed)
    checked = 0
    for text in (fuzz.random_scene(seed), fuzz.mesh_scene(seed, str(tmp_path / "soup.obj"))):
        s = pkg.Scene(text, ROOT, device=None)
        ref = orc.OracleScene(text, base_dir=ROOT)
        d = host.rth_scene_desc(s._h).contents
        for li in range(d.n_leaves):
            L = d.leaves[li]
            if L.type not in (2, 6):  # RT_PLANE, RT_TRIANGLE
                continue
            ln = d.triangles[L.tri].normal if L.type == 6 else L.plane_normal
            got = (ctypes.c_double * 3)()
            assert fn(d.xforms[L.xform].full_inv, ln, got) == 0
            want, ty = (ctypes.c_double * 3)(), ctypes.c_int()
            at = (ctypes.c_double * 3)(*rng.uniform(-20, 20, 3))  # the point does not matter for these shapes
            assert olib.orc_leaf_normal_at(ref._h, li, at, 0.25, 0.5, want, ctypes.byref(ty)) == 0
            a, b = np.array(got[:]), np.array(want[:])
            assert a.tobytes() == b.tobytes() or (np.isnan(a) == np.isnan(b)).all() and (a[~np.isnan(a)] == b[~np.isnan(b)]).all(), (li, a, b)
            checked += 1
    assert checked > 0
