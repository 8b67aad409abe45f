"""GPU parity on seeded random scenes: every primitive type under random affine transforms, nested
groups, CSG pairs, patterns, mirrors and glass, rendered by the HIP path (through the C ABI) and by the
CPU oracle from the same JSON text.  The shipped scenes place shapes tidily; these do not: shapes
overlap, groups are rotated and sheared into each other, cameras sit inside shapes, some scales are
huge or tiny, some cameras look exactly along an axis (the |d| < 1e-4 rules of cubes and group boxes).

Bar: hit / miss mask and primary hit leaf bit-exact, f64 colours within 1e-9 (NaN where the oracle has
NaN), 8-bit image never more than 1 LSB off, primary and shadow ray counts equal.
"""
import json
import math
import os

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu

PRIMS = ["sphere", "cube", "cylinder", "bounded_cylinder", "capped_cylinder", "bounded_cone", "bounded_dn_cone",
         "capped_cone", "capped_dn_cone", "plane"]
PATTERNS = ["stripe", "ring", "checker", "gradient", "point", "identity"]


def _mat(rng, spread, lo=0.3, hi=1.6, wild=False):
    """Row-major 4x4 = translation * Rx * Ry * Rz * (shear) * scaling."""
    def rot(axis, a):
        c, s = math.cos(a), math.sin(a)
        m = np.eye(4)
        i, j = [(1, 2), (0, 2), (0, 1)][axis]
        m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
        return m
    s = np.eye(4)
    scale = rng.uniform(lo, hi, 3)
    if wild:
        k = rng.integers(0, 4)
        if k == 0:
            scale = scale * 150.0   # beyond the size at which cubes and cones leave the hierarchy
        elif k == 1:
            scale[rng.integers(0, 3)] = 0.01
        elif k == 2:
            scale[rng.integers(0, 3)] *= -1.0  # mirrored
    s[0, 0], s[1, 1], s[2, 2] = scale
    sh = np.eye(4)
    if rng.random() < 0.2:
        sh[rng.integers(0, 3), (rng.integers(0, 2) + 1) % 3] = rng.uniform(-0.7, 0.7)
        sh[np.diag_indices(4)] = 1.0
    t = np.eye(4)
    t[:3, 3] = rng.uniform(-spread, spread, 3) * np.array([1.0, 0.5, 1.0]) + np.array([0.0, spread * 0.4, 0.0])
    r = np.eye(4)
    if rng.random() < 0.8:
        r = rot(0, rng.uniform(0, 2 * math.pi)) @ rot(1, rng.uniform(0, 2 * math.pi)) @ rot(2, rng.uniform(0, 2 * math.pi))
    return [float(v) for v in (t @ r @ sh @ s).reshape(-1)]


def _material(rng):
    m = {"color": [float(v) for v in rng.uniform(0.05, 1.0, 3)]}
    if rng.random() < 0.4:
        p = {"ty": str(rng.choice(PATTERNS)), "primary_color": [float(v) for v in rng.uniform(0, 1, 3)],
             "secondary_color": [float(v) for v in rng.uniform(0, 1, 3)]}
        if rng.random() < 0.7:
            p["transform"] = _mat(rng, 1.0, 0.2, 1.5)
        m["pattern"] = p
    if rng.random() < 0.5:
        m.update(ambient=float(rng.uniform(0, 0.4)), diffuse=float(rng.uniform(0.2, 1.0)),
                 specular=float(rng.uniform(0, 1.0)), shininess=float(rng.choice([1.0, 10.0, 37.5, 200.0, 300.0, 1500.0])))
    if rng.random() < 0.3:
        m["reflective"] = float(rng.uniform(0.1, 0.9))
    if rng.random() < 0.25:
        m["transparency"] = float(rng.uniform(0.3, 0.95))
        m["refractive_index"] = float(rng.choice([1.0, 1.1, 1.33, 1.5, 2.417]))
    return m


def _primitive(rng, spread, wild_ok=True):
    s = {"ty": str(rng.choice(PRIMS)), "material": _material(rng)}
    if rng.random() < 0.92:
        s["transform"] = _mat(rng, spread, wild=wild_ok and rng.random() < 0.06)
    return s


def _shape(rng, spread, depth):
    u = rng.random()
    if depth < 3 and u < 0.22:
        g = {"ty": "group", "children": [_shape(rng, spread * 0.6, depth + 1) for _ in range(rng.integers(0, 5))]}
        if rng.random() < 0.8:
            g["transform"] = _mat(rng, spread * 0.5, 0.5, 1.4)
        return g
    if u < 0.30:
        kids = []
        for _ in range(2):
            k = _primitive(rng, 0.8, wild_ok=False)
            k["ty"] = str(rng.choice(["sphere", "cube", "capped_cylinder"]))
            kids.append(k)
        c = {"ty": str(rng.choice(["union", "intersection", "difference"])), "children": kids}
        if rng.random() < 0.8:
            c["transform"] = _mat(rng, spread, 0.6, 1.5)
        return c
    return _primitive(rng, spread)


def random_scene(seed, width=80, height=56):
    rng = np.random.default_rng(0xF0221 + seed)
    spread = float(rng.choice([2.0, 4.0, 8.0]))
    shapes = [_shape(rng, spread, 0) for _ in range(rng.integers(1, 22))]
    if rng.random() < 0.7:
        shapes.append({"ty": "plane", "material": _material(rng)})
    a = rng.uniform(0, 2 * math.pi)
    dist = spread * rng.uniform(0.3, 2.5)  # sometimes in the middle of the shapes
    cam_from = [dist * math.cos(a), float(rng.uniform(0.2, spread)), dist * math.sin(a)]
    cam_to = [float(v) for v in rng.uniform(-1, 1, 3) + np.array([0.0, 1.0, 0.0])]
    cam_up = [0.0, 1.0, 0.0]
    if rng.random() < 0.15:  # looking exactly along an axis: direction components that are exactly 0 and below 1e-4
        cam_from, cam_to = [0.0, 1.0, -dist - 1.0], [0.0, 1.0, 0.0]
    light = [float(v) for v in rng.uniform(-spread, spread, 3) * np.array([1.5, 1.0, 1.5]) + np.array([0, spread * 1.5, 0])]
    if rng.random() < 0.1:
        light[1] = -abs(light[1])  # below the floor
    return json.dumps({
        "canvas_width": width, "canvas_height": height, "field_of_view": float(rng.uniform(0.5, 1.6)),
        "camera_from": cam_from, "camera_to": cam_to, "camera_up": cam_up,
        "light": {"intensity": [float(v) for v in rng.uniform(0.5, 1.2, 3)], "position": light},
        "shapes": shapes})


def compare(out, stats, want, label):
    n = want["hit_leaf"].size
    mask_diff = int(((out["hit_leaf"] >= 0) != (want["hit_leaf"] >= 0)).sum())
    leaf_diff = int((out["hit_leaf"] != want["hit_leaf"]).sum())
    nan_cpu, nan_gpu = np.isnan(want["rgb64"]), np.isnan(out["rgb64"])
    nan_diff = int((nan_cpu != nan_gpu).sum())
    ok = ~(nan_cpu | nan_gpu)
    d = np.abs(np.where(ok, out["rgb64"], 0.0) - np.where(ok, want["rgb64"], 0.0))
    scale = np.maximum(1.0, np.abs(np.where(ok, want["rgb64"], 0.0)))
    err = float((d / scale).max()) if n else 0.0
    d8 = np.abs(out["rgb8"].astype(int) - orc.quantise(want["rgb64"]).astype(int))
    worse, off1 = int((d8 > 1).sum()), int((d8 == 1).sum())
    print("%s: pixels=%d hits=%d mask_diff=%d leaf_diff=%d nan=%d nan_diff=%d lsb1=%d lsb>1=%d err=%.2e rays gpu=%s cpu=%s" % (
        label, n, int((want["hit_leaf"] >= 0).sum()), mask_diff, leaf_diff, int(nan_cpu.sum()), nan_diff, off1, worse, err,
        [stats.primary, stats.shadow, stats.reflect, stats.refract], want["counts"]))
    assert mask_diff == 0, "hit/miss mask differs"
    assert leaf_diff == 0, "primary hit leaf differs"
    assert nan_diff == 0, "NaN colours differ"
    assert err <= 1e-9, "f64 colours differ beyond rounding"
    assert worse == 0 and off1 <= max(1, n // 1000), "8-bit image differs"
    assert [stats.primary, stats.shadow] == want["counts"][:2]
    assert abs(stats.reflect - want["counts"][2]) <= max(2, want["counts"][2] * 1e-3)
    assert abs(stats.refract - want["counts"][3]) <= max(2, want["counts"][3] * 1e-3)


# RT_FUZZ_SEEDS, RT_FUZZ_CROWDS, RT_FUZZ_MESHES widen the sweeps (1500 / 80 / 400 seeds were run once: all equal)
@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_FUZZ_SEEDS", "120"))))
def test_random_scene_matches_oracle(pkg, root, seed):
    text = random_scene(seed)
    cpu = orc.OracleScene(text, base_dir=root)
    try:
        gpu = pkg.Scene(text, base_dir=root, device=0)
    except pkg.RtError as e:
        # the only scene the device may refuse: a CSG tree with more possible hits per ray than it holds
        assert "CSG" in str(e) or "csg" in str(e), str(e)
        pytest.skip("refused: %s" % e)
    out, stats = gpu.render()
    want = cpu.render(jobs=8)
    compare(out, stats, want, "seed %d" % seed)


def crowd_scene(seed, width=64, height=48):
    """Hundreds of small primitives, half of them cones: deep hierarchies, and enough bounded cones for
    the direction grid of their parallel-ray branch (more than 32) to be built and used."""
    rng = np.random.default_rng(0xC20D + seed)
    n = int(rng.integers(120, 420))
    spread = float(rng.choice([6.0, 12.0]))
    shapes = []
    for _ in range(n):
        ty = str(rng.choice(["bounded_cone", "capped_cone", "bounded_dn_cone", "capped_dn_cone", "sphere", "capped_cylinder",
                             "bounded_cylinder", "cube"]))
        s = {"ty": ty, "material": _material(rng), "transform": _mat(rng, spread, 0.15, 0.9)}
        if rng.random() < 0.1:
            s = {"ty": "group", "transform": _mat(rng, 1.0, 0.7, 1.3), "children": [s, _primitive(rng, spread, wild_ok=False)]}
        shapes.append(s)
    shapes.append({"ty": "plane", "material": _material(rng)})
    a = rng.uniform(0, 2 * math.pi)
    return json.dumps({
        "canvas_width": width, "canvas_height": height, "field_of_view": float(rng.uniform(0.7, 1.3)),
        "camera_from": [1.8 * spread * math.cos(a), float(rng.uniform(1.0, spread)), 1.8 * spread * math.sin(a)],
        "camera_to": [0.0, spread * 0.3, 0.0], "camera_up": [0.0, 1.0, 0.0],
        "light": {"intensity": [1.0, 1.0, 1.0], "position": [float(v) for v in rng.uniform(-spread, spread, 3) + np.array([0, 2 * spread, 0])]},
        "shapes": shapes})


@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_FUZZ_CROWDS", "24"))))
def test_random_crowd_matches_oracle(pkg, root, seed):
    text = crowd_scene(seed)
    gpu = pkg.Scene(text, base_dir=root, device=0)
    cpu = orc.OracleScene(text, base_dir=root)
    out, stats = gpu.render()
    compare(out, stats, cpu.render(jobs=8), "crowd %d" % seed)


def mesh_scene(seed, obj_path, width=72, height=48):
    """A random triangle soup (flat and smooth faces, polygons, OBJ groups) written as an OBJ file and
    loaded as a model: under its own transform, inside a transformed group (so the triangles sit three
    transform levels down and two of them are not the identity), next to a second, untransformed copy."""
    rng = np.random.default_rng(0x0B7 + seed)
    nv = int(rng.integers(12, 120))
    verts = rng.uniform(-1.5, 1.5, (nv, 3))
    normals = rng.normal(size=(nv, 3))
    lines = ["# random soup %d" % seed]
    lines += ["v %r %r %r" % tuple(float(c) for c in v) for v in verts]
    smooth = rng.random() < 0.6
    if smooth:
        lines += ["vn %r %r %r" % tuple(float(c) for c in v) for v in normals]
    for f in range(int(rng.integers(10, 160))):
        if rng.random() < 0.08:
            lines.append("g part%d" % int(rng.integers(0, 4)))
        k = 3 if rng.random() < 0.8 else int(rng.integers(4, 6))  # polygons are fanned (src/obj.rs)
        idx = rng.choice(nv, k, replace=False) + 1
        lines.append("f " + " ".join(("%d//%d" % (i, i)) if smooth else str(i) for i in idx))
    with open(obj_path, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    model = {"ty": "model", "path": obj_path, "transform": _mat(rng, 1.0, 0.6, 1.4)}
    shapes = [{"ty": "group", "transform": _mat(rng, 1.5, 0.7, 1.3), "children": [model, _primitive(rng, 2.0, wild_ok=False)]},
              {"ty": "model", "path": obj_path},
              {"ty": "plane", "transform": [1, 0, 0, 0, 0, 1, 0, -2.5, 0, 0, 1, 0, 0, 0, 0, 1], "material": _material(rng)}]
    a = rng.uniform(0, 2 * math.pi)
    return json.dumps({
        "canvas_width": width, "canvas_height": height, "field_of_view": 1.0,
        "camera_from": [7 * math.cos(a), float(rng.uniform(0.5, 4)), 7 * math.sin(a)], "camera_to": [0.0, 0.5, 0.0],
        "camera_up": [0.0, 1.0, 0.0],
        "light": {"intensity": [1.0, 1.0, 1.0], "position": [float(v) for v in rng.uniform(-6, 6, 3) + np.array([0, 9, 0])]},
        "shapes": shapes})


@pytest.mark.parametrize("seed", range(int(os.environ.get("RT_FUZZ_MESHES", "12"))))
def test_random_mesh_matches_oracle(pkg, root, seed, tmp_path):
    text = mesh_scene(seed, str(tmp_path / "soup.obj"))
    gpu = pkg.Scene(text, base_dir=root, device=0)
    cpu = orc.OracleScene(text, base_dir=root)
    assert gpu.info()["leaves"] == cpu.info()["leaves"]
    out, stats = gpu.render()
    compare(out, stats, cpu.render(jobs=8), "mesh %d" % seed)


@pytest.mark.parametrize("seed", range(4))
def test_caller_rays_through_a_crowd(pkg, root, seed):
    """The per-ray entry points on a crowd scene with rays a camera never produces: directions that are
    not unit vectors (the cone direction grid is built for unit directions; the others take the list),
    origins inside the crowd, axis-parallel directions.  Hit leaf and t must be identical, colours to
    rounding, shadow answers equal."""
    text = crowd_scene(100 + seed)
    gpu = pkg.Scene(text, base_dir=root, device=0)
    cpu = orc.OracleScene(text, base_dir=root)
    rng = np.random.default_rng(seed)
    n = 3000
    rays = np.empty((n, 6))
    rays[:, :3] = rng.uniform(-10, 10, (n, 3)) * np.array([1.0, 0.5, 1.0]) + np.array([0.0, 4.0, 0.0])
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 3] *= rng.uniform(0.05, 30.0, (n // 3, 1))       # not unit length
    axis = rng.integers(0, 3, n // 10)
    d[n // 3: n // 3 + n // 10] = np.eye(3)[axis] * rng.choice([-1.0, 1.0], (n // 10, 1))  # along an axis
    rays[:, 3:] = d
    t, leaf = gpu.intersect_hit(rays)
    wt, wleaf = cpu.intersect_hit(rays)
    assert np.array_equal(leaf, wleaf)
    assert np.array_equal(t[wleaf >= 0], wt[wleaf >= 0])
    rgb, leaf2, _ = gpu.color_at(rays)
    want, wleaf2, _ = cpu.color_at(rays)
    assert np.array_equal(leaf2, wleaf2)
    # unbounded multipliers (a direction of length 30 reflected five times) overflow in both alike
    fin = np.isfinite(rgb) & np.isfinite(want)
    assert np.array_equal(rgb[~fin], want[~fin], equal_nan=True)
    err = np.abs(np.where(fin, rgb, 0) - np.where(fin, want, 0)) / np.maximum(1.0, np.abs(np.where(fin, want, 0)))
    print("crowd rays %d: hits %d max err %.2e" % (seed, int((wleaf >= 0).sum()), err.max()))
    assert err.max() <= 1e-9
    pts = rays[:, :3]
    assert np.array_equal(gpu.is_shadowed(pts), cpu.is_shadowed(pts))
