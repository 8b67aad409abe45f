"""The acceleration structure built on the GPU (bvh_device.hip: binned surface-area splits top down, collapse
to the 4-wide layout) against the CPU oracle (the reference's algorithm: every child of every group scanned,
src/shape.rs:1213-1219) and against the hierarchy the host builds (binned SAH): all are conservative over the
same shapes, so every image, hit leaf and ray count must be the same."""
import json
import os
import sys

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, GOLDEN)


class builder:
    """Chooses the builder of the scenes created inside the block (the library reads the environment per scene)."""

    def __init__(self, device, min_items=2):
        self.env = {"RT_AMD_DEVICE_BVH": "1" if device else "0", "RT_AMD_DEVICE_BVH_MIN": str(min_items)}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def both(pkg, text, size=None):
    out = []
    for device in (False, True):
        with builder(device):
            s = pkg.Scene(text, ROOT, device=0)
        if size:
            s.set_canvas(*size)
        out.append((s, s.render(), s.footprint()))
    return out


def equals_oracle(render, text, size=None):
    """The parity bar of tests/test_gpu_parity.py against the oracle's render of the same scene text."""
    out, st = render
    ref = orc.OracleScene(text, ROOT)
    if size:
        ref.set_canvas(*size)
    want = ref.render(jobs=8)
    assert np.array_equal(out["hit_leaf"], want["hit_leaf"])
    assert np.array_equal(np.isnan(out["rgb64"]), np.isnan(want["rgb64"]))
    assert np.nanmax(np.abs(out["rgb64"] - want["rgb64"]), initial=0.0) <= 1e-9
    assert np.array_equal(out["rgb8"], orc.quantise(want["rgb64"]))
    assert [st.primary, st.shadow, st.reflect, st.refract] == want["counts"]
    return ref


def same(a, b):
    (ra, sa), (rb, sb) = a, b
    assert np.array_equal(ra["hit_leaf"], rb["hit_leaf"])
    assert np.array_equal(ra["rgb64"], rb["rgb64"], equal_nan=True) and np.array_equal(ra["rgb8"], rb["rgb8"])
    assert (sa.primary, sa.shadow, sa.reflect, sa.refract) == (sb.primary, sb.shadow, sb.reflect, sb.refract)


@pytest.mark.parametrize("name", ["example", "egg", "glass", "pattern", "teapot", "allprims", "csg", "heightfield", "primitives"])
def test_images_do_not_depend_on_the_builder(pkg, name, tmp_path):
    size = None
    if name in ("allprims", "csg"):
        text = open(os.path.join(GOLDEN, name + ".json")).read()
    elif name in ("heightfield", "primitives"):
        import make_golden
        text = open(make_golden.synthetic_scene(name, str(tmp_path))).read()
    else:
        text = open(os.path.join(ROOT, "scenes", name + ".json")).read()
        size = (128, 72)
    (host, r_host, f_host), (dev, r_dev, f_dev) = both(pkg, text, size)
    ref = equals_oracle(r_dev, text, size)  # the device-built hierarchy against the reference's algorithm
    same(r_host, r_dev)
    assert f_host["device_builds"] == 0
    if host.info()["leaves"] >= 4:  # a subtree of two items or more exists
        assert f_dev["device_builds"] >= 1
    rays = np.random.default_rng(4).uniform(-1, 1, (500, 6))
    rays[:, :3] *= 8.0
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    ca, la, _ = host.color_at(rays)
    cb, lb, _ = dev.color_at(rays)
    assert np.array_equal(la, lb) and np.array_equal(ca, cb, equal_nan=True)
    assert np.array_equal(host.is_shadowed(rays[:, :3]), dev.is_shadowed(rays[:, :3]))
    want, wleaf, counts = ref.color_at(rays)
    assert np.array_equal(lb, wleaf) and np.array_equal(np.isnan(cb), np.isnan(want))
    assert np.nanmax(np.abs(cb - want), initial=0.0) <= 1e-9
    t, leaf = dev.intersect_hit(rays)
    wt, wl = ref.intersect_hit(rays)
    assert np.array_equal(leaf, wl) and np.array_equal(t[leaf >= 0], wt[leaf >= 0])
    assert np.array_equal(dev.is_shadowed(rays[:, :3]), ref.is_shadowed(rays[:, :3]))


def test_crowd_with_groups_built_on_the_device(pkg, tmp_path):
    """Several hundred primitives, half of them in nested transformed groups: gates as items of the device build."""
    import scenegen
    path, n = scenegen.primitives(str(tmp_path), count=400, width=128, height=72, name="crowd")
    scene = json.load(open(path))
    shapes = scene["shapes"]
    groups = []
    for k in range(0, 200, 25):
        inner = {"ty": "group", "transform": scenegen.translation(0.5, 0.0, -0.5), "children": shapes[k + 10:k + 25]}
        groups.append({"ty": "group", "children": shapes[k:k + 10] + [inner]})
    scene["shapes"] = groups + shapes[200:]
    (host, r_host, _), (dev, r_dev, f_dev) = both(pkg, json.dumps(scene))
    equals_oracle(r_dev, json.dumps(scene))
    same(r_host, r_dev)
    assert f_dev["device_builds"] >= 9


def test_shapes_the_splits_cannot_separate(pkg):
    """Where no bin boundary separates a node's items -- many shapes at one place, shapes strung along one line --
    the device builder halves the stretch instead; and two items are the smallest tree there is."""
    def ball(x, y, z, r):
        return {"ty": "sphere", "transform": [r, 0, 0, x, 0, r, 0, y, 0, 0, r, z, 0, 0, 0, 1],
                "material": {"color": [0.3 + 0.1 * (int(x * 7) % 5), 0.5, 0.7]}}

    base = json.load(open(os.path.join(ROOT, "scenes", "example.json")))
    base.update(canvas_width=96, canvas_height=54, camera_from=[0.0, 3.0, -14.0], camera_to=[0.0, 0.5, 0.0])
    crowds = {
        "300 spheres at one place, 5 elsewhere": [ball(0.0, 1.0, 0.0, 1.0)] * 300 + [ball(-4.0 + 2.0 * k, 0.0, 2.0, 0.6) for k in range(5)],
        "400 spheres along a line, touching": [ball(-6.0 + 0.03 * k, 0.0, 0.0, 0.5) for k in range(400)],
        "two spheres": [ball(-1.0, 0.0, 0.0, 1.0), ball(1.5, 0.5, 1.0, 1.0)],
        "three spheres": [ball(-2.0, 0.0, 0.0, 1.0), ball(0.5, 0.5, 1.0, 1.0), ball(3.0, 0.0, 0.0, 1.0)],
    }
    for what, shapes in crowds.items():
        text = json.dumps(dict(base, shapes=shapes))
        (host, r_host, f_host), (dev, r_dev, f_dev) = both(pkg, text)
        assert f_dev["device_builds"] >= 1 and f_host["device_builds"] == 0, what
        same(r_host, r_dev)
        equals_oracle(r_dev, text)


def test_a_million_triangles_build_in_milliseconds(pkg, tmp_path):
    import scenegen
    path, tris = scenegen.heightfield(str(tmp_path), cells=708, name="hf_device")
    assert tris == 1002528
    text = open(path).read()
    with builder(True, min_items=2048):
        pkg.Scene(open(os.path.join(ROOT, "scenes", "teapot.json")).read(), ROOT, device=0)  # loads the builder's kernels
        dev = pkg.Scene(text, ROOT, device=0)
    f = dev.footprint()
    print("device build of 1M triangles: %.2f ms in %d build(s); whole acceleration-structure phase %.1f ms" % (
        f["device_build_seconds"] * 1e3, f["device_builds"], f["build_seconds"] * 1e3))
    assert f["device_builds"] >= 1
    assert f["device_build_seconds"] < 0.040  # 20 ms measured; the host builder takes 180 ms
    dev.set_canvas(960, 540)
    r_dev = dev.render(rgb64=False)
    with builder(False):
        host = pkg.Scene(text, ROOT, device=0)
    host.set_canvas(960, 540)
    r_host = host.render(rgb64=False)
    assert np.array_equal(r_dev[0]["hit_leaf"], r_host[0]["hit_leaf"]) and np.array_equal(r_dev[0]["rgb8"], r_host[0]["rgb8"])
    assert r_dev[1].shadow == r_host[1].shadow
    # same splitting rule in both builders: a ray visits as many nodes through either hierarchy
    c_dev, c_host = dev.render(rgb64=False, counters=True)[1], host.render(rgb64=False, counters=True)[1]
    assert c_dev.node_visits <= 1.03 * c_host.node_visits and c_dev.tri_tests <= 1.03 * c_host.tri_tests
    # with nothing asked for, a subtree this large goes to the GPU and a small one stays on the host
    for k in ("RT_AMD_DEVICE_BVH", "RT_AMD_DEVICE_BVH_MIN"):
        assert k not in os.environ
    assert pkg.Scene(text, ROOT, device=0).footprint()["device_builds"] >= 1
    assert pkg.Scene(open(os.path.join(ROOT, "scenes", "teapot.json")).read(), ROOT, device=0).footprint()["device_builds"] == 0


def test_a_device_built_scene_is_updated_in_place(pkg):
    import scenegen
    text = open(os.path.join(GOLDEN, "allprims.json")).read()
    scene = json.loads(text)
    with builder(True):
        moving = pkg.Scene(text, ROOT, device=0)
    moving.render()
    m = scenegen.matmul(scenegen.translation(-3.0, 0.5, 0.5), scenegen.rot_y(0.9))
    scene["shapes"][1]["transform"] = list(m)
    moving.set_transform([1], m)
    moving.update()
    with builder(False):
        fresh = pkg.Scene(json.dumps(scene), ROOT, device=0)
    moved = moving.render()
    equals_oracle(moved, json.dumps(scene))
    same(moved, fresh.render())


class kernels:
    """Chooses the kernel variants of the scenes created inside the block (read from the environment per scene)."""

    def __init__(self, large):
        v = "1" if large else "0"
        self.env = {"RT_AMD_UNIFORM_NODES": v, "RT_AMD_TRI_PREFETCH": v}

    __enter__ = builder.__enter__
    __exit__ = builder.__exit__


@pytest.mark.parametrize("name", ["teapot", "allprims", "heightfield"])
def test_images_do_not_depend_on_the_kernel_variant(pkg, name, tmp_path):
    """Large scenes get kernels that read wave-uniform nodes through the scalar cache and fetch the next
    triangle of a run while one is tested: forced on small scenes, every image is the same bit for bit."""
    size = None
    if name == "allprims":
        text = open(os.path.join(GOLDEN, name + ".json")).read()
    elif name == "heightfield":
        import make_golden
        text = open(make_golden.synthetic_scene(name, str(tmp_path))).read()
    else:
        text = open(os.path.join(ROOT, "scenes", name + ".json")).read()
        size = (160, 90)
    out = []
    for large in (False, True):
        with kernels(large):
            s = pkg.Scene(text, ROOT, device=0)
        if size:
            s.set_canvas(*size)
        out.append(s.render())
    equals_oracle(out[1], text, size)
    same(out[0], out[1])
