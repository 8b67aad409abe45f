"""GPU results against the committed golden fixtures (tests/golden/, written by make_golden.py from the oracle)."""
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
F64_TOL = 1e-9  # f64 colours: same operations in the same order; only pow / libm differ in the last bits


def check(out, stats, g, n_pix):
    assert np.array_equal(out["hit_leaf"], g["hit_leaf"])  # the primary hit of every pixel: bit exact
    assert np.abs(out["rgb64"] - g["rgb64"]).max() <= F64_TOL
    assert np.array_equal(out["rgb8"], orc.quantise(g["rgb64"]))  # the 8-bit image: no pixel off, not even by 1 LSB
    assert [stats.primary, stats.shadow, stats.reflect, stats.refract] == [int(c) for c in g["counts"]]


@pytest.mark.parametrize("name", ["example", "egg", "glass", "pattern", "teapot"])
def test_shipped_scene_against_golden(pkg, name):
    g = np.load(os.path.join(GOLDEN, "scene_%s.npz" % name))
    s = pkg.Scene.from_file(os.path.join(ROOT, "scenes", name + ".json"), base_dir=ROOT, device=0)
    w, h = int(g["size"][0]), int(g["size"][1])
    s.set_canvas(w, h)
    out, st = s.render()
    check(out, st, g, w * h)


@pytest.mark.parametrize("name", ["heightfield", "primitives"])
def test_synthetic_configuration_against_golden(pkg, name, tmp_path):
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden
    g = np.load(os.path.join(GOLDEN, "synthetic_%s.npz" % name))
    s = pkg.Scene.from_file(make_golden.synthetic_scene(name, str(tmp_path)), base_dir=ROOT, device=0)
    out, st = s.render()
    check(out, st, g, out["hit_leaf"].size)


def test_primitive_field_sweep_frames_against_golden(pkg, tmp_path):
    """Two more camera positions of the 60-frame sweep over the tiny primitive field."""
    import sys
    sys.path.insert(0, GOLDEN)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden
    import scenegen
    g = np.load(os.path.join(GOLDEN, "synthetic_primitives_sweep.npz"))
    s = pkg.Scene.from_file(make_golden.synthetic_scene("primitives", str(tmp_path)), base_dir=ROOT, device=0)
    w, h = make_golden.SYNTHETIC["primitives"]["width"], make_golden.SYNTHETIC["primitives"]["height"]
    assert [int(k) for k in g["frames"]] == list(make_golden.SWEEP_FRAMES)
    for k in make_golden.SWEEP_FRAMES:
        cam_from, cam_to = scenegen.sweep_camera(k)
        s.set_camera(w, h, 1.0472, cam_from, cam_to, [0.0, 1.0, 0.0])
        out, st = s.render()
        check(out, st, {"hit_leaf": g["hit_leaf_%d" % k], "rgb64": g["rgb64_%d" % k], "counts": g["counts_%d" % k]}, w * h)


@pytest.mark.parametrize("name", ["allprims", "csg"])
def test_every_primitive_pattern_group_and_operator_against_golden(pkg, name):
    """allprims: every primitive, pattern, nested groups, nested glass.  csg: union / intersection /
    difference, nested, under and above groups, with a cone (no finite box), glass, equal operands."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        s = pkg.Scene(f.read(), ROOT, device=0)
    out, st = s.render()
    check(out, st, g, out["hit_leaf"].size)
    rgb, leaf, st2 = s.color_at(g["rays"])
    assert np.array_equal(leaf, g["ray_leaf"])
    assert np.abs(rgb - g["ray_rgb"]).max() <= F64_TOL
    assert [st2.primary, st2.shadow, st2.reflect, st2.refract] == [int(c) for c in g["ray_counts"]]
    t, leaf2 = s.intersect_hit(g["rays"])
    assert np.array_equal(leaf2, g["ray_leaf"])
    hit = leaf2 >= 0
    assert np.isnan(t[~hit]).all()
    if name == "allprims":
        # identical object-space arithmetic for top-level shapes: the distance itself is bit exact
        assert np.array_equal(t[hit], g["ray_t"][hit])
    else:
        # below operators with their own transform the ray goes through one composite matrix here and
        # through one matrix per level in the reference: equal to rounding
        assert np.allclose(t[hit], g["ray_t"][hit], rtol=1e-12, atol=1e-12)


def test_remaining_limits_the_generations(pkg):
    g = np.load(os.path.join(GOLDEN, "allprims.npz"))
    with open(os.path.join(GOLDEN, "allprims.json")) as f:
        text = f.read()
    s = pkg.Scene(text, ROOT, device=0)
    ref = orc.OracleScene(text, ROOT)
    rays = g["rays"][:200]
    for remaining in (0, 1, 3):
        rgb, leaf, st = s.color_at(rays, remaining=remaining)
        want, wleaf, counts = ref.color_at(rays, remaining=remaining)
        assert np.array_equal(leaf, wleaf)
        assert np.abs(rgb - want).max() <= F64_TOL
        assert [st.primary, st.shadow, st.reflect, st.refract] == counts
        assert st.generations <= remaining + 1


def test_remaining_beyond_the_renderer_s_depth(pkg):
    """World::color_at takes any `remaining` (src/world.rs:249); the renderer passes 5.  Rays between two facing
    mirrors and through nested glass with remaining 6, 9 and 40 against the oracle, then the depth of 5 again on the
    same scene (the tables grown for the deep calls serve the shallow one)."""
    import json
    with open(os.path.join(GOLDEN, "allprims.json")) as f:
        scene = json.load(f)
    mirror = {"color": [0.9, 0.9, 1.0], "reflective": 0.9, "diffuse": 0.2, "specular": 0.1}
    scene["shapes"] += [
        {"ty": "cube", "transform": [0.05, 0, 0, -4.0, 0, 3, 0, 14.0, 0, 0, 3, 0, 0, 0, 0, 1], "material": mirror},
        {"ty": "cube", "transform": [0.05, 0, 0, 4.0, 0, 3, 0, 14.0, 0, 0, 3, 0, 0, 0, 0, 1], "material": mirror},
    ]
    text = json.dumps(scene)
    s = pkg.Scene(text, ROOT, device=0)
    ref = orc.OracleScene(text, ROOT)
    g = np.load(os.path.join(GOLDEN, "allprims.npz"))
    rng = np.random.default_rng(2)
    n = 120
    o = np.stack([rng.uniform(-3.0, 3.0, n), rng.uniform(12.0, 16.0, n), rng.uniform(-2.0, 2.0, n)], axis=1)
    d = np.stack([rng.choice([-1.0, 1.0], n), rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n)], axis=1)
    hall = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], axis=1)
    rays = np.concatenate([hall, g["rays"][:150]])
    deepest = 0
    for remaining in (6, 9, 40, 5):
        rgb, leaf, st = s.color_at(rays, remaining=remaining)
        want, wleaf, counts = ref.color_at(rays, remaining=remaining)
        assert np.array_equal(leaf, wleaf)
        assert np.abs(rgb - want).max() <= F64_TOL
        assert [st.primary, st.shadow, st.reflect, st.refract] == counts
        assert st.generations <= remaining + 1
        deepest = max(deepest, st.generations)
    assert deepest > 20  # the hall of mirrors really goes that deep
    with pytest.raises(pkg.RtError) as e:
        s.color_at(rays[:2], remaining=5000)
    assert e.value.status == -2
