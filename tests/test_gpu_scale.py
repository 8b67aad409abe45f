"""Full-size and many-primitive checks: properties that need no full CPU render, plus oracle
comparisons on a pixel subsample (the reference algorithm is O(primitives) per ray)."""
import json
import os
import sys

import numpy as np
import pytest

import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scenegen  # noqa: E402


def subsample_check(gpu_out, ref, stride, x0=0, y0=0, tol=1e-9):
    want = ref.render(jobs=8, x0=x0, xstep=stride, y0=y0, ystep=stride)
    sl = (slice(y0, None, stride), slice(x0, None, stride))
    assert np.array_equal(gpu_out["hit_leaf"][sl], want["hit_leaf"][sl])
    assert np.abs(gpu_out["rgb64"][sl] - want["rgb64"][sl]).max() <= tol
    q = orc.quantise(want["rgb64"][sl])
    assert np.array_equal(gpu_out["rgb8"][sl], q)
    return want


def test_teapot_1080p(pkg):
    """The headline configuration at full size."""
    path = os.path.join(ROOT, "scenes", "teapot.json")
    s = pkg.Scene.from_file(path, base_dir=ROOT, device=0)
    s.set_canvas(1920, 1080)
    out, st = s.render()
    assert st.primary == 1920 * 1080 and st.reflect == 0 and st.refract == 0
    assert st.shadow == int((out["hit_leaf"] >= 0).sum())  # one shadow query per shaded hit
    assert np.array_equal(out["rgb8"], orc.quantise(out["rgb64"]))  # on-device quantisation == Canvas::save's
    again, _ = s.render()
    assert np.array_equal(again["rgb64"], out["rgb64"])  # idempotent, no run-to-run noise
    acc = np.zeros_like(out["rgb64"])
    shadow = 0
    for part in range(8):  # the 8-GPU partition, part by part on one GPU
        o, stp = s.render(part=part, n_parts=8, rgb8=False, hit_leaf=False)
        acc += o["rgb64"]
        shadow += stp.shadow
    assert np.array_equal(acc, out["rgb64"]) and shadow == st.shadow
    ref = orc.OracleScene.from_file(path, base_dir=ROOT)
    ref.set_canvas(1920, 1080)
    subsample_check(out, ref, 2, x0=1, y0=1)  # 960 x 540 pixels of the same frame through the oracle (bench.py checks all of them)


def test_glass_4k_properties(pkg):
    path = os.path.join(ROOT, "scenes", "glass.json")
    s = pkg.Scene.from_file(path, base_dir=ROOT, device=0)  # 3840 x 2160 as shipped
    out, st = s.render()
    assert st.primary == 3840 * 2160
    assert st.reflect == st.refract  # the only spawning material is the glass sphere: one of each per hit...
    assert st.shadow == st.primary - int((out["hit_leaf"] < 0).sum()) + (st.shadow - int((out["hit_leaf"] >= 0).sum()))
    assert np.array_equal(out["rgb8"], orc.quantise(out["rgb64"]))
    ref = orc.OracleScene.from_file(path, base_dir=ROOT)
    subsample_check(out, ref, 1)  # every pixel of the 4K frame


@pytest.fixture(scope="module")
def gen_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("gen"))


def test_heightfield_mesh_small_against_oracle(pkg, gen_dir):
    """The 1M-triangle configuration's generator at 24 x 24 cells (1152 triangles), full image vs the oracle."""
    path, tris = scenegen.heightfield(gen_dir, cells=24, width=160, height=90, name="hf_small")
    s = pkg.Scene.from_file(path, base_dir=ROOT, device=0)
    assert s.info()["triangles"] == tris == 1152
    out, st = s.render()
    ref = orc.OracleScene.from_file(path, base_dir=ROOT)
    want = ref.render(jobs=8)
    assert np.array_equal(out["hit_leaf"], want["hit_leaf"])
    assert np.abs(out["rgb64"] - want["rgb64"]).max() <= 1e-9
    assert [st.primary, st.shadow, st.reflect, st.refract] == want["counts"]


def test_primitive_field_small_against_oracle(pkg, gen_dir):
    """The 100k-primitive configuration's generator at 300 primitives, two camera positions of the sweep."""
    path, n = scenegen.primitives(gen_dir, count=300, width=160, height=90, name="prims_small")
    s = pkg.Scene.from_file(path, base_dir=ROOT, device=0)
    ref = orc.OracleScene.from_file(path, base_dir=ROOT)
    assert s.info()["leaves"] == n == ref.info()["leaves"]
    for frame in (0, 17):
        cam_from, cam_to = scenegen.sweep_camera(frame)
        s.set_camera(160, 90, 1.0472, cam_from, cam_to, [0.0, 1.0, 0.0])
        ref.set_camera(160, 90, 1.0472, cam_from, cam_to, [0.0, 1.0, 0.0])
        out, st = s.render()
        want = ref.render(jobs=8)
        assert np.array_equal(out["hit_leaf"], want["hit_leaf"])
        assert np.abs(out["rgb64"] - want["rgb64"]).max() <= 1e-9
        assert np.array_equal(out["rgb8"], orc.quantise(want["rgb64"]))
        assert [st.primary, st.shadow, st.reflect, st.refract] == want["counts"]


def test_heightfield_mesh_full_size(pkg, gen_dir):
    """1 002 528 triangles at 3840 x 2160: properties, and a sparse subsample through the oracle."""
    path, tris = scenegen.heightfield(gen_dir, cells=708, name="hf_full")
    assert tris == 1002528
    s = pkg.Scene.from_file(path, base_dir=ROOT, device=0)
    out, st = s.render()
    assert st.primary == 3840 * 2160 and st.reflect == 0 and st.refract == 0
    assert st.shadow == int((out["hit_leaf"] >= 0).sum())
    assert np.array_equal(out["rgb8"], orc.quantise(out["rgb64"]))
    again, _ = s.render(rgb8=False, hit_leaf=False)
    assert np.array_equal(again["rgb64"], out["rgb64"])
    ref = orc.OracleScene.from_file(path, base_dir=ROOT)
    subsample_check(out, ref, 120, x0=100, y0=60)  # 32 x 18 pixels: each costs two scans of a million triangles


def test_primitive_field_full_size(pkg, gen_dir):
    """100 000 top-level primitives at 7680 x 4320, one frame of the sweep."""
    path, n = scenegen.primitives(gen_dir, count=100000, name="prims_full")
    s = pkg.Scene.from_file(path, base_dir=ROOT, device=0)
    assert s.info()["leaves"] == n
    out, st = s.render(rgb64=False)
    assert st.primary == 7680 * 4320
    assert st.shadow >= int((out["hit_leaf"] >= 0).sum())
    assert st.refract == 0 and st.reflect > 0
    rays = s.ray_for_pixel(np.array([[x, y] for y in range(100, 4320, 240) for x in range(100, 7680, 320)], np.uint32))  # 432 pixels
    ref = orc.OracleScene.from_file(path, base_dir=ROOT)
    rgb, leaf, _ = s.color_at(rays)
    want, wleaf, _ = ref.color_at(rays)
    assert np.array_equal(leaf, wleaf)
    assert np.abs(rgb - want).max() <= 1e-9
