"""Frames of a scene that spawns rays, only enqueued (rt_render with RT_FLAG_ASYNC).  Every generation of such a
frame is launched at once on the queue lengths of the last verified frame; whether the queues had room is known
when the frame's counters are back, and the library looks at them when the frame's image is downloaded, when
rt_scene_wait is called, or before the next blocking call.  A frame that did not fit is rendered again, exactly,
into the same buffers first.  Whatever the host reads must equal the blocking render (World::color_at per pixel,
src/world.rs:249-261; the reference renders one frame at a time, src/parallel.rs:183-204)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H = 320, 180
NEAR = ([0.0, 1.5, -18.0], [0.0, 1.0, 0.0])      # the scene's own view: spheres of glass and mirrors fill the frame
AWAY = ([0.0, 30.0, -18.0], [0.0, 60.0, -18.5])  # from above the scene into the sky: no hit, no secondary ray


def frames_on_the_device(pkg, count):
    amd, _ = pkg.bindings.libraries()
    amd.rt_device_alloc.argtypes = [C.c_int, C.c_uint64, C.POINTER(C.c_void_p)]
    ptrs = []
    for _ in range(count):
        p = C.c_void_p()
        assert amd.rt_device_alloc(0, W * H * 3, C.byref(p)) == 0
        ptrs.append(p.value)
    return amd, ptrs


def blocking(scene, view):
    scene.set_camera(W, H, 1.0472, view[0], view[1], [0.0, 1.0, 0.0])
    out, st = scene.render(rgb64=False, hit_leaf=False)
    return out["rgb8"], st


def test_enqueued_frames_equal_the_blocking_render_whether_they_fit_or_not(pkg, root):
    scene = pkg.Scene.from_file(os.path.join(root, "scenes", "glass.json"), base_dir=root, device=0)
    want_near, st_near = blocking(scene, NEAR)
    want_away, st_away = blocking(scene, AWAY)
    # with no secondary ray in the frame the queue lengths are all zero: a frame enqueued on them that spawns a
    # single ray cannot fit, so stretch 2 below is certain to take the path that renders again
    assert st_away.reflect + st_away.refract == 0 and st_near.reflect + st_near.refract > 10000
    amd, dev = frames_on_the_device(pkg, 4)
    ring = pkg.bindings.HostFrameRing(W, H, 4)
    amd.rt_scene_rebuilds.restype = C.c_int
    try:
        # 1. a stretch of frames from one camera: the first is exact (no queue lengths yet), the rest only enqueued
        blocking(scene, NEAR)
        scene.wait()
        for k in range(6):
            scene.render_device(rgb8_ptr=dev[k % 4], wait=False)
            if k >= 1:  # the image of the frame before: verified here, while this one runs
                scene.download(dev[(k - 1) % 4], ring.address + ((k - 1) % 4) * ring.frame_bytes, ring.frame_bytes)
        scene.download(dev[5 % 4], ring.address + (5 % 4) * ring.frame_bytes, ring.frame_bytes)
        total = scene.wait()
        assert total.primary == 6 * W * H
        assert (total.shadow, total.reflect, total.refract) == (6 * st_near.shadow, 6 * st_near.reflect, 6 * st_near.refract)
        for slot in range(4):
            assert np.array_equal(ring.frame(slot), want_near), slot
        # 2. the queue lengths of the view into the sky, then frames of the near view enqueued on them: none of them fits
        blocking(scene, AWAY)
        scene.set_camera(W, H, 1.0472, NEAR[0], NEAR[1], [0.0, 1.0, 0.0])
        for k in range(3):
            scene.render_device(rgb8_ptr=dev[k], wait=False)
        for k in range(3):
            ring.frame(k)[:] = 0
            scene.download(dev[k], ring.address + k * ring.frame_bytes, ring.frame_bytes)
        total = scene.wait()
        assert (total.primary, total.reflect, total.refract) == (3 * W * H, 3 * st_near.reflect, 3 * st_near.refract)
        for k in range(3):
            assert np.array_equal(ring.frame(k), want_near), k
        # 3. and back: room to spare, and rt_scene_wait alone verifies what nobody downloaded
        scene.set_camera(W, H, 1.0472, AWAY[0], AWAY[1], [0.0, 1.0, 0.0])
        for k in range(5):
            scene.render_device(rgb8_ptr=dev[k % 4], wait=False)
        total = scene.wait()
        assert (total.primary, total.reflect, total.refract) == (5 * W * H, 5 * st_away.reflect, 5 * st_away.refract)
        scene.download(dev[0], ring.address, ring.frame_bytes)
        scene.wait()
        assert np.array_equal(ring.frame(0), want_away)
        # 4. a blocking frame after enqueued ones sees their counts accounted for, not mixed into its own
        scene.render_device(rgb8_ptr=dev[1], wait=False)
        out, st = scene.render(rgb64=False, hit_leaf=False)
        assert np.array_equal(out["rgb8"], want_away) and (st.reflect, st.refract) == (st_away.reflect, st_away.refract)
        assert scene.wait().primary == W * H
    finally:
        ring.close()
        amd.rt_device_free.argtypes = [C.c_int, C.c_void_p]
        for p in dev:
            amd.rt_device_free(0, C.c_void_p(p))
