"""CPU tests of the oracle: the reference's known-answer vectors and the committed golden fixtures."""
import json
import os
import subprocess

import numpy as np
import pytest

import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_reference_known_answer_vectors():
    """oracle/kat_tests restates the reference's #[test] vectors for the render path."""
    exe = os.path.join(ROOT, "oracle", "_build", "kat_tests")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout[-2000:]
    assert "0 failures" in p.stdout


@pytest.mark.parametrize("name", ["example", "egg", "glass", "pattern", "teapot"])
def test_scene_matches_golden(name):
    g = np.load(os.path.join(GOLDEN, "scene_%s.npz" % name))
    s = orc.OracleScene.from_file(os.path.join(ROOT, "scenes", name + ".json"), base_dir=ROOT)
    s.set_canvas(int(g["size"][0]), int(g["size"][1]))
    r = s.render(jobs=4)
    assert np.array_equal(r["hit_leaf"], g["hit_leaf"])
    assert np.array_equal(r["rgb64"], g["rgb64"])  # same compiler flags, same libm: bit for bit
    assert r["counts"] == [int(c) for c in g["counts"]]


@pytest.mark.parametrize("name", ["heightfield", "primitives"])
def test_synthetic_generators_match_golden(name, tmp_path):
    """The seeded generators of the two synthetic configurations are deterministic: a tiny instance
    regenerated here renders to the committed fixture."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden
    g = np.load(os.path.join(GOLDEN, "synthetic_%s.npz" % name))
    s = orc.OracleScene.from_file(make_golden.synthetic_scene(name, str(tmp_path)), base_dir=ROOT)
    r = s.render(jobs=4)
    assert np.array_equal(r["hit_leaf"], g["hit_leaf"])
    assert np.array_equal(r["rgb64"], g["rgb64"])
    assert r["counts"] == [int(c) for c in g["counts"]]


@pytest.mark.parametrize("name", ["allprims", "csg"])
def test_allprims_matches_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        s = orc.OracleScene(f.read(), ROOT)
    r = s.render(jobs=4)
    assert np.array_equal(r["hit_leaf"], g["hit_leaf"])
    assert np.allclose(r["rgb64"], g["rgb64"], rtol=0, atol=1e-12)
    rgb, leaf, counts = s.color_at(g["rays"])
    assert np.array_equal(leaf, g["ray_leaf"])
    assert np.allclose(rgb, g["ray_rgb"], rtol=0, atol=1e-12)
    t, _ = s.intersect_hit(g["rays"])
    assert np.array_equal(np.isnan(t), np.isnan(g["ray_t"]))
    assert np.array_equal(t[~np.isnan(t)], g["ray_t"][~np.isnan(t)])


def test_jobs_do_not_change_the_image():
    s = orc.OracleScene.from_file(os.path.join(ROOT, "scenes", "glass.json"), base_dir=ROOT)
    s.set_canvas(40, 24)
    a, b = s.render(jobs=1), s.render(jobs=5)
    assert np.array_equal(a["rgb64"], b["rgb64"]) and a["counts"] == b["counts"]


def test_strided_render_is_a_subset():
    s = orc.OracleScene.from_file(os.path.join(ROOT, "scenes", "pattern.json"), base_dir=ROOT)
    s.set_canvas(40, 24)
    full = s.render(jobs=2)
    part = s.render(jobs=2, x0=1, xstep=4, y0=2, ystep=3)
    assert np.array_equal(part["rgb64"][2::3, 1::4], full["rgb64"][2::3, 1::4])
    assert part["counts"][0] == len(range(1, 40, 4)) * len(range(2, 24, 3))


def test_scene_errors_surface_as_exceptions():
    with pytest.raises(orc.OracleError):
        orc.OracleScene('{"canvas_width": 4}', ROOT)
    base = json.load(open(os.path.join(ROOT, "scenes", "example.json")))
    bad = dict(base, shapes=[{"ty": "dodecahedron"}])
    with pytest.raises(orc.OracleError, match="Unrecognized shape type"):
        orc.OracleScene(json.dumps(bad), ROOT)
    singular = dict(base, shapes=[{"ty": "sphere", "transform": [1, 0, 0]}])
    with pytest.raises(orc.OracleError, match="inverse"):
        orc.OracleScene(json.dumps(singular), ROOT)
    missing = dict(base, shapes=[{"ty": "model", "path": "models/none.obj"}])
    with pytest.raises(orc.OracleError, match="Failed to open"):
        orc.OracleScene(json.dumps(missing), ROOT)


def test_empty_and_ragged_inputs():
    base = json.load(open(os.path.join(ROOT, "scenes", "example.json")))
    empty = dict(base, shapes=[], canvas_width=5, canvas_height=3)
    r = orc.OracleScene(json.dumps(empty), ROOT).render(jobs=2)
    assert not r["rgb64"].any() and (r["hit_leaf"] == -1).all() and r["counts"] == [15, 0, 0, 0]
    ragged = dict(base, canvas_width=7, canvas_height=5, camera_up=[0, 1], shapes=[
        {"ty": "sphere", "material": {"color": [1]}}, {"ty": "group"}, {"ty": "group", "children": []}])
    r = orc.OracleScene(json.dumps(ragged), ROOT).render(jobs=1)
    assert r["rgb64"][..., 1:].max() <= r["rgb64"][..., 0].max()  # colour [1] is zero-filled to (1, 0, 0)
