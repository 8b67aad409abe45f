"""rt_render_frames: up to eight frames of one scene, each through its own camera, in one call.  For a scene that
spawns no rays the frames are one kernel launch; every frame must equal, byte for byte, the frame rt_render makes
for the same camera and part (the reference renders one frame per run, src/main.rs:71).  The frame loop uses such
launches for the parts of a frame."""
import copy
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 233, 141  # no multiple of a tile or a band


def cameras_around(scene, n):
    """n views of the teapot from different places (RtCamera structs)."""
    cams = []
    for k in range(n):
        scene.set_camera(W, H, 1.0472 - 0.05 * k, [1.5 * k - 4.0, 1.5 + 0.4 * k, -10.0 + k], [0.0, 1.0, 0.0], [0.0, 1.0, 0.0])
        cams.append(copy.deepcopy(scene.camera()))
    return cams


def alone(scene, torch, cam_setter, part, parts, rows):
    cam_setter()
    frame = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    scene.render_device(rgb8_ptr=frame.data_ptr(), part=part, n_parts=parts, bands=parts > 1, band_rows=rows)
    return frame.cpu().numpy()


@pytest.mark.parametrize("part,parts,rows", [(0, 1, 32), (1, 2, 8), (3, 8, 8), (2, 3, 16), (0, 5, 32)])
def test_frames_of_one_launch_equal_frames_rendered_alone(pkg, root, part, parts, rows):
    import torch
    scene = pkg.Scene.from_file(os.path.join(root, "scenes", "teapot.json"), base_dir=root, device=0)
    n = 8
    cams = cameras_around(scene, n)
    want = []
    for k in range(n):
        def put(k=k):
            scene._camera = cams[k]
        want.append(alone(scene, torch, put, part, parts, rows))
    assert any(not np.array_equal(want[0], w) for w in want[1:])  # the views differ
    for count in (8, 5, 2, 1):
        frames = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0") for _ in range(count)]
        scene.wait()
        scene.render_frames_device(cams[:count], [f.data_ptr() for f in frames], part=part, n_parts=parts, bands=parts > 1, band_rows=rows)
        st = scene.wait()
        assert st.kernel_launches == 1 and st.generations == count
        own = int((want[0].any(axis=2)).sum())  # lit pixels of the part: a lower bound of what it owns
        assert st.primary % count == 0 and st.primary // count >= own
        for k in range(count):
            assert np.array_equal(frames[k].cpu().numpy(), want[k]), "frame %d of %d" % (k, count)


@pytest.mark.parametrize("kind", ["large mesh", "csg"])
def test_the_other_kernel_variants_render_several_frames_too(pkg, root, tmp_path, kind):
    """A mesh large enough for the kernels that read wave-uniform nodes through the scalar cache and fetch a run's
    next triangle ahead, and a scene with CSG operators (the general kernels): the same launches of several frames."""
    import json
    import sys
    import torch
    sys.path.insert(0, os.path.join(root, "tools"))
    import scenegen
    if kind == "large mesh":
        path, tris = scenegen.heightfield(str(tmp_path), cells=256, width=W, height=H, name="hf_frames")
        scene = pkg.Scene.from_file(path, base_dir=root, device=0)
        assert tris >= 100000
        views = [([0.3 * k, 2.0 + 0.2 * k, -4.0 + 0.3 * k], [0.0, 0.0, 0.0]) for k in range(3)]
    else:
        base = json.load(open(os.path.join(root, "scenes", "example.json")))
        z05 = scenegen.translation(0.0, 0.0, 0.5)
        shapes = [{"ty": "difference", "children": [{"ty": "cube"}, {"ty": "sphere", "transform": z05}]},
                  {"ty": "union", "transform": scenegen.translation(2.5, 0.0, 0.0), "children": [{"ty": "sphere"}, {"ty": "cylinder", "transform": z05}]},
                  {"ty": "plane", "transform": scenegen.translation(0.0, -1.0, 0.0)}]
        scene = pkg.Scene(json.dumps(dict(base, shapes=shapes)), root, device=0)
        views = [([1.0 * k - 1.0, 1.5 + 0.5 * k, -6.0], [0.5, 0.0, 0.0]) for k in range(3)]
    cams = []
    for cam_from, cam_to in views:
        scene.set_camera(W, H, 1.0472, cam_from, cam_to, [0.0, 1.0, 0.0])
        cams.append(copy.deepcopy(scene.camera()))
    for part, parts in ((0, 1), (1, 3)):
        want = []
        for k in range(3):
            def put(k=k):
                scene._camera = cams[k]
            want.append(alone(scene, torch, put, part, parts, 8))
        assert want[0].any() and not np.array_equal(want[0], want[1])
        frames = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0") for _ in range(3)]
        scene.wait()
        scene.render_frames_device(cams, [f.data_ptr() for f in frames], part=part, n_parts=parts, bands=parts > 1, band_rows=8)
        st = scene.wait()
        assert st.kernel_launches == 1 and st.generations == 3
        for k in range(3):
            assert np.array_equal(frames[k].cpu().numpy(), want[k]), "%s: frame %d, part %d of %d" % (kind, k, part, parts)


def test_scenes_that_spawn_rays_get_their_frames_one_after_the_other(pkg, root):
    import torch
    scene = pkg.Scene.from_file(os.path.join(root, "scenes", "glass.json"), base_dir=root, device=0)
    scene.set_canvas(W, H)
    cam = copy.deepcopy(scene.camera())
    want, _ = scene.render(rgb64=False, hit_leaf=False)
    frames = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0") for _ in range(3)]
    scene.render_frames_device([cam] * 3, [f.data_ptr() for f in frames])
    st = scene.wait()
    assert st.primary == 3 * W * H and st.generations > 3 and st.reflect > 0
    for f in frames:
        assert np.array_equal(f.cpu().numpy(), want["rgb8"])


def test_refusals(pkg, root):
    import torch
    scene = pkg.Scene.from_file(os.path.join(root, "scenes", "teapot.json"), base_dir=root, device=0)
    scene.set_canvas(W, H)
    amd, _ = pkg.libraries()
    cam = scene.camera()
    frame = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    RtCamera = type(cam)

    def call(n, flags, cams=None, ptrs=None):
        cams = cams or [cam] * max(n, 1)
        arr = (RtCamera * len(cams))(*cams)
        p = (C.c_void_p * len(cams))(*(ptrs or [C.c_void_p(frame.data_ptr())] * len(cams)))
        return amd.rt_render_frames(scene._dev(), arr, n, 0, 1, flags, p)

    assert call(0, 2) == -1
    assert call(9, 2) == -1
    assert call(2, 0) == -1  # not asynchronous
    other = copy.deepcopy(cam)
    other.hsize += 1
    assert call(2, 2, cams=[cam, other]) == -1  # two canvas sizes
    assert call(2, 2, ptrs=[C.c_void_p(frame.data_ptr()), C.c_void_p(None)]) == -1
    assert call(2, 2) == 0
    scene.wait()


@pytest.mark.parametrize("parts,depth,per_launch", [(2, 8, 2), (4, 16, 4), (8, 12, 4), (3, 9, 3), (4, 3, 1), (1, 8, 2), (1, 12, 4)])
def test_the_frame_loop_launches_several_frames_of_a_part_at_once(pkg, root, parts, depth, per_launch):
    path = os.path.join(root, "scenes", "teapot.json")
    whole = pkg.Scene.from_file(path, base_dir=root, device=0)
    whole.set_canvas(200, 150)
    want = whole.render(rgb64=False, hit_leaf=False)[0]["rgb8"]
    host = pkg.Scene.from_file(path, base_dir=root, device=None)
    host.set_canvas(200, 150)
    loop = host.frame_loop([0] * parts, depth=depth)
    for warm, frames in ((3, 11), (0, 8), (1, 1)):  # stretches that do not divide by the launch size
        for slot in range(depth):
            loop.host_frame(slot)[:] = 0
        seconds, stats = loop.run(warm, frames, True)
        assert loop.frames_per_launch == per_launch
        assert sum(s.primary for s in stats) == frames * 200 * 150
        if per_launch > 1 and frames >= per_launch:
            assert sum(s.kernel_launches for s in stats) < frames * parts
        for slot in range(min(depth, warm + frames)):
            assert np.array_equal(loop.host_frame(slot), want), "parts=%d slot=%d" % (parts, slot)
    loop.close()
