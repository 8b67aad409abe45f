"""Shapes that move between frames (Shape::set_transform, src/shape.rs:592-656): the scene is updated in
place (rt_scene_update: new boxes for the same hierarchy, refitted on the GPU, no rebuild) and must render
exactly what a scene freshly built from the moved world renders, and what the oracle renders."""
import copy
import json
import math
import os
import sys

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from scenegen import matmul, rot_x, rot_y, rot_z, scaling, translation  # noqa: E402


def shape_at(scene, path):
    s = scene["shapes"][path[0]]
    for k in path[1:]:
        s = s["children"][k]
    return s


def check_moves(pkg, scene, moves, size=None, oracle=True):
    """moves: list of frames, each a list of (path, matrix).  One GPU scene is updated frame by frame."""
    scene = copy.deepcopy(scene)
    if size:
        scene["canvas_width"], scene["canvas_height"] = size
    moving = pkg.Scene(json.dumps(scene), ROOT, device=0)
    moving.render()
    for frame in moves:
        for path, m in frame:
            shape_at(scene, path)["transform"] = list(m)
            moving.set_transform(path, m)
        moving.update()
        got, st = moving.render()
        text = json.dumps(scene)
        fresh = pkg.Scene(text, ROOT, device=0)
        want, st0 = fresh.render()
        assert np.array_equal(got["hit_leaf"], want["hit_leaf"])
        assert np.array_equal(got["rgb64"], want["rgb64"], equal_nan=True) and np.array_equal(got["rgb8"], want["rgb8"])
        assert (st.primary, st.shadow, st.reflect, st.refract) == (st0.primary, st0.shadow, st0.reflect, st0.refract)
        if oracle:
            ref = orc.OracleScene(text, ROOT).render(jobs=8)
            assert np.array_equal(got["hit_leaf"], ref["hit_leaf"])
            assert np.abs(got["rgb64"] - ref["rgb64"]).max() <= 1e-9
            assert np.array_equal(got["rgb8"], orc.quantise(ref["rgb64"]))
            assert [st.primary, st.shadow, st.reflect, st.refract] == ref["counts"]


def test_shapes_and_groups_of_every_kind_move(pkg):
    """allprims: quadrics of every type, patterns, nested glass, a transformed group holding a transformed group."""
    with open(os.path.join(GOLDEN, "allprims.json")) as f:
        scene = json.load(f)
    moves = [
        [([1], matmul(translation(-3.5, 0.4, 1.0), matmul(rot_y(1.1), scaling(0.7, 1.2, 0.5)))),   # the cube
         ([5], matmul(translation(2.0, -1, 2.0), matmul(rot_z(0.3), scaling(0.9, 1.4, 0.9))))],   # a bounded cone
        [([11], matmul(translation(-0.5, 2.2, 3.0), rot_y(-0.7))),                                 # the outer group
         ([11, 1], matmul(translation(1.0, 0.3, 0), scaling(0.8, 0.5, 0.8))),                      # the group inside it
         ([11, 1, 0], matmul(rot_z(1.0), rot_x(0.4)))],                                            # a cube two groups down
        [([9], matmul(translation(0.0, 0.7, -1.5), scaling(1.0, 1.0, 1.0))),                       # the glass sphere, off its bubble
         ([0], translation(0, -1.3, 0))],                                                          # the floor plane
    ]
    check_moves(pkg, scene, moves)


def test_the_teapot_mesh_moves(pkg):
    with open(os.path.join(ROOT, "scenes", "teapot.json")) as f:
        scene = json.load(f)
    model = [i for i, s in enumerate(scene["shapes"]) if s["ty"] == "model"][0]
    base = scene["shapes"][model].get("transform") or matmul(translation(0, 0, 0), scaling(1, 1, 1))
    moves = [[([model], matmul(translation(0.8, 0.3, -0.5), matmul(rot_y(0.6 * k), base)))] for k in (1, 2)]
    check_moves(pkg, scene, moves, size=(96, 54))


def test_a_field_of_primitives_moves(pkg, tmp_path):
    """More than 32 cones: their direction index is rebuilt with the new matrices."""
    import scenegen
    path, n = scenegen.primitives(str(tmp_path), count=240, width=96, height=54, name="moving")
    with open(path) as f:
        scene = json.load(f)
    rng = np.random.default_rng(21)
    frame = []
    for i in rng.choice(np.arange(1, n), 60, replace=False):
        t = rng.uniform(-30, 30, 3)
        t[1] = rng.uniform(0, 12)
        a = rng.uniform(0, math.tau, 3)
        sc = rng.uniform(0.3, 1.2, 3)
        frame.append(([int(i)], matmul(translation(*t), matmul(rot_x(a[0]), matmul(rot_y(a[1]), matmul(rot_z(a[2]), scaling(*sc)))))))
    check_moves(pkg, scene, [frame, frame[:20]])


def rebuilds(pkg, scene):
    import ctypes as C
    amd, _ = pkg.bindings.libraries()
    amd.rt_scene_rebuilds.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    n = C.c_uint64()
    assert amd.rt_scene_rebuilds(scene._dev(), C.byref(n)) == 0
    return n.value


def check_against_oracle(got, st, text):
    ref = orc.OracleScene(text, ROOT).render(jobs=8)
    assert np.array_equal(got["hit_leaf"], ref["hit_leaf"])
    assert np.abs(got["rgb64"] - ref["rgb64"]).max() <= 1e-9
    assert np.array_equal(got["rgb8"], orc.quantise(ref["rgb64"]))
    assert [st.primary, st.shadow, st.reflect, st.refract] == ref["counts"]


def test_updates_the_hierarchy_cannot_follow_build_it_again(pkg):
    """Shape::set_transform works on any shape (src/shape.rs:592-656).  Where a refit is not enough -- a shape that
    crosses between the hierarchy and the shapes tested for every ray, a transform chain that changes its length,
    a scene with CSG operators -- rt_scene_update builds the acceleration structure again; the scene keeps its
    handle and renders what the oracle renders for the moved world."""
    with open(os.path.join(GOLDEN, "allprims.json")) as f:
        scene = json.load(f)
    s = pkg.Scene(json.dumps(scene), ROOT, device=0)
    s.render()
    assert rebuilds(pkg, s) == 0
    big = scaling(150, 150, 150)  # a cube that large is tested for every ray, not through the hierarchy
    s.set_transform([1], big)
    s.update()
    assert rebuilds(pkg, s) == 1
    scene["shapes"][1]["transform"] = list(big)
    got, st = s.render()
    check_against_oracle(got, st, json.dumps(scene))
    small = matmul(translation(-3.5, 0.4, 1.0), scaling(0.7, 1.2, 0.5))  # and back into the hierarchy
    s.set_transform([1], small)
    s.update()
    assert rebuilds(pkg, s) == 2
    scene["shapes"][1]["transform"] = list(small)
    got, st = s.render()
    check_against_oracle(got, st, json.dumps(scene))
    # a group two levels down gets a transform of its own where it had none: its shapes' chains grow by one matrix
    s.set_transform([11, 1], translation(0.0, 0.0, 0.0))
    s.update()
    s.set_transform([11, 1], matmul(translation(1.0, 0.3, 0), scaling(0.8, 0.5, 0.8)))
    s.update()
    scene["shapes"][11]["children"][1]["transform"] = list(matmul(translation(1.0, 0.3, 0), scaling(0.8, 0.5, 0.8)))
    got, st = s.render()
    check_against_oracle(got, st, json.dumps(scene))
    # scenes with CSG operators
    with open(os.path.join(GOLDEN, "csg.json")) as f:
        csg_scene = json.load(f)
    csg = pkg.Scene(json.dumps(csg_scene), ROOT, device=0)
    csg.render()
    moved = matmul(translation(0.3, 1.0, -0.2), rot_y(0.5))
    base = csg_scene["shapes"][1].get("transform")
    m = matmul(moved, base) if base else moved
    csg.set_transform([1], m)
    csg.update()
    assert rebuilds(pkg, csg) == 1
    csg_scene["shapes"][1]["transform"] = list(m)
    got, st = csg.render()
    check_against_oracle(got, st, json.dumps(csg_scene))
