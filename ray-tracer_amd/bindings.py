"""ctypes bindings of the C ABI (include/rt_amd.h) and of the host library.

Names follow the reference: ``Scene`` bundles a World and a Camera as
src/scene.rs:20-23 does, ``Scene.parallel_render`` is src/parallel.rs:183,
``Scene.color_at`` / ``ray_for_pixel`` / ``is_shadowed`` / ``intersect_hit``
are the batch forms of src/world.rs:249, src/camera.rs:76, src/world.rs:123 and
src/world.rs:107 + src/intersect.rs:71.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBDIR = os.path.join(_HERE, "lib")

RT_FLAG_COUNTERS = 1
RT_FLAG_ASYNC = 2
RT_FLAG_COMPACT = 4
RT_FLAG_BANDS = 8
RT_FLAG_BANDS_16 = 16
RT_FLAG_BANDS_8 = 32


class RtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rt status %d: %s" % (status, message))
        self.status = status
        self.message = message


class RtCamera(C.Structure):
    _fields_ = [
        ("hsize", C.c_uint32),
        ("vsize", C.c_uint32),
        ("half_width", C.c_double),
        ("half_height", C.c_double),
        ("pixel_size", C.c_double),
        ("inv", C.c_double * 16),
    ]


class RtStats(C.Structure):
    _fields_ = [
        ("primary", C.c_uint64),
        ("shadow", C.c_uint64),
        ("reflect", C.c_uint64),
        ("refract", C.c_uint64),
        ("render_ms", C.c_double),
        ("kernel_ms", C.c_double),
        ("kernel_launches", C.c_uint64),
        ("generations", C.c_uint64),
        ("node_visits", C.c_uint64),
        ("tri_tests", C.c_uint64),
        ("quadric_tests", C.c_uint64),
        ("gate_tests", C.c_uint64),
        ("shadow_node_visits", C.c_uint64),
        ("shadow_tri_tests", C.c_uint64),
        ("shadow_quadric_tests", C.c_uint64),
        ("shadow_gate_tests", C.c_uint64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}

    @property
    def rays(self):
        return self.primary + self.shadow + self.reflect + self.refract


_libs = None


def libraries():
    """Loads (once) and returns (librt_amd, librt_host).  Raises if they are not built."""
    global _libs
    if _libs is not None:
        return _libs
    # RT_AMD_LIB / RT_HOST_LIB: file names of alternative builds in lib/ (kernel tuning experiments)
    names = (os.environ.get("RT_AMD_LIB", "librt_amd.so"), os.environ.get("RT_HOST_LIB", "librt_host.so"))
    paths = [os.path.join(_LIBDIR, n) for n in names]
    for p in paths:
        if not os.path.exists(p):
            raise RtError(-3, "%s is missing: run `make -C %s` (there is no CPU fallback)" % (p, _HERE))
    amd = C.CDLL(paths[0], mode=C.RTLD_GLOBAL)
    host = C.CDLL(paths[1], mode=C.RTLD_GLOBAL)
    vp, u64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int, C.c_uint32
    dp = C.POINTER(C.c_double)
    amd.rt_last_error.restype = C.c_char_p
    amd.rt_device_count.restype = i32
    amd.rt_scene_footprint.argtypes = [vp, C.POINTER(u64), dp]
    amd.rt_scene_build_info.argtypes = [vp, C.POINTER(u64), dp]
    amd.rt_render.argtypes = [vp, C.POINTER(RtCamera), i32, i32, u32, vp, vp, vp, C.POINTER(RtStats)]
    amd.rt_scene_wait.argtypes = [vp, C.POINTER(RtStats)]
    amd.rt_render_frames.argtypes = [vp, C.POINTER(RtCamera), i32, i32, i32, u32, C.POINTER(vp)]
    amd.rt_scene_stream.argtypes = [vp, C.POINTER(vp)]
    amd.rt_scene_copy_stream.argtypes = [vp, C.POINTER(vp)]
    amd.rt_download.argtypes = [vp, vp, vp, u64, vp]
    amd.rt_download_2d.argtypes = [vp, vp, vp, u64, u64, u64, vp]
    amd.rt_host_alloc.argtypes = [u64]
    amd.rt_host_alloc.restype = vp
    amd.rt_host_free.argtypes = [vp]
    amd.rt_host_free.restype = None
    amd.rt_host_register.argtypes = [vp, u64]
    amd.rt_host_unregister.argtypes = [vp]
    amd.rt_render_host.argtypes = [vp, C.POINTER(RtCamera), i32, i32, u32, vp, vp, vp, C.POINTER(RtStats)]
    amd.rt_ray_for_pixel.argtypes = [vp, C.POINTER(RtCamera), u64, vp, vp]
    amd.rt_color_at.argtypes = [vp, u64, vp, i32, u32, vp, vp, C.POINTER(RtStats)]
    amd.rt_intersect_hit.argtypes = [vp, u64, vp, vp, vp]
    amd.rt_is_shadowed.argtypes = [vp, u64, vp, vp]
    amd.rt_ppm_encode.argtypes = [i32, vp, u64, u64, vp, u64, C.POINTER(u64)]
    amd.rt_ppm_encode_host.argtypes = [i32, vp, u64, u64, vp, u64, C.POINTER(u64)]
    amd.rt_ppm_encode_static.argtypes = [i32, vp, u64, u64, vp, u64, C.POINTER(u64)]
    amd.rt_ppm_static_max_pixels.restype = u64
    amd.rt_ppm_max_bytes.argtypes = [u64, u64]
    amd.rt_ppm_max_bytes.restype = u64
    host.rth_last_error.restype = C.c_char_p
    host.rth_scene_from_json.argtypes = [C.c_char_p, C.c_char_p, i32, C.POINTER(vp)]
    host.rth_scene_free.argtypes = [vp]
    host.rth_scene_device.argtypes = [vp]
    host.rth_scene_device.restype = vp
    host.rth_scene_info.argtypes = [vp, C.POINTER(u64)]
    host.rth_scene_camera.argtypes = [vp, C.POINTER(RtCamera)]
    host.rth_scene_set_canvas.argtypes = [vp, u64, u64]
    host.rth_scene_set_camera.argtypes = [vp, u64, u64, C.c_double, dp, dp, dp]
    host.rth_scene_set_transform.argtypes = [vp, C.POINTER(u64), u64, dp]
    host.rth_scene_update.argtypes = [vp]
    host.rth_parallel_render.argtypes = [vp, u64, C.c_char_p]
    host.rth_canvas_new.argtypes = [u64, u64, C.POINTER(vp)]
    host.rth_render_canvas.argtypes = [vp, u64, C.POINTER(vp)]
    host.rth_canvas_free.argtypes = [vp]
    host.rth_canvas_size.argtypes = [vp, C.POINTER(u64), C.POINTER(u64)]
    host.rth_canvas_write_pixel.argtypes = [vp, u64, u64, dp]
    host.rth_canvas_read_pixel.argtypes = [vp, u64, u64, dp]
    host.rth_canvas_pixels.argtypes = [vp]
    host.rth_canvas_pixels.restype = dp
    host.rth_canvas_save.argtypes = [vp, C.c_char_p]
    host.rth_frame_loop_new.argtypes = [vp, u64, C.POINTER(C.c_int32), C.c_int32, C.POINTER(vp)]
    host.rth_frame_loop_new_shared.argtypes = [vp, u64, C.POINTER(C.c_int32), C.c_int32, u64, u64, vp, C.POINTER(vp)]
    host.rth_frame_loop_free.argtypes = [vp]
    host.rth_frame_loop_run.argtypes = [vp, u64, u64, C.c_int32, dp, C.POINTER(RtStats)]
    host.rth_frame_loop_blocking.argtypes = [vp, u64, C.c_int32, C.POINTER(RtStats)]
    host.rth_frame_loop_host_frame.argtypes = [vp, C.c_int32]
    host.rth_frame_loop_host_frame.restype = vp
    host.rth_frame_loop_info.argtypes = [vp, C.POINTER(u64), dp]
    amd.rt_scene_clone.argtypes = [vp, i32, C.POINTER(vp)]
    amd.rt_copy_fence.argtypes = [vp, u32]
    amd.rt_copy_wait.argtypes = [vp, u32]
    amd.rt_device_alloc.argtypes = [i32, u64, C.POINTER(vp)]
    amd.rt_device_free.argtypes = [i32, vp]
    host.rth_ppm_encode.argtypes = [u64, u64, vp, vp, u64]
    host.rth_ppm_encode.restype = u64
    _libs = (amd, host)
    return _libs


def device_count():
    amd, _ = libraries()
    n = amd.rt_device_count()
    if n < 0:
        raise RtError(n, amd.rt_last_error().decode())
    return n


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def ppm_encode(rgb8):
    """Canvas::save's text (src/canvas.rs:47-94) for an (H, W, 3) uint8 image."""
    _, host = libraries()
    img = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _c = img.shape
    size = host.rth_ppm_encode(w, h, _ptr(img), None, 0)
    buf = C.create_string_buffer(size)
    host.rth_ppm_encode(w, h, _ptr(img), buf, size)
    return buf.raw


def ppm_max_bytes(width, height):
    """Size of the device buffer ppm_encode_device needs for a width x height frame."""
    amd, _ = libraries()
    return int(amd.rt_ppm_max_bytes(width, height))


def ppm_encode_device(rgb8_ptr, width, height, out_ptr, capacity, device=0, static=False):
    """Canvas::save's text (src/canvas.rs:47-94) written on the GPU: rgb8_ptr and out_ptr are device
    addresses.  Returns the length of the file in out_ptr.  static=True: rt_ppm_encode_static, which
    works in the library's fixed workspace and allocates nothing."""
    amd, _ = libraries()
    n = C.c_uint64()
    fn = amd.rt_ppm_encode_static if static else amd.rt_ppm_encode
    rc = fn(device, rgb8_ptr, width, height, out_ptr, capacity, C.byref(n))
    if rc != 0:
        raise RtError(rc, amd.rt_last_error().decode())
    return int(n.value)


def ppm_encode_via_device(rgb8, device=0):
    """Canvas::save's text for an (H, W, 3) uint8 host image, formatted on the GPU."""
    amd, _ = libraries()
    img = np.ascontiguousarray(rgb8, dtype=np.uint8)
    h, w, _c = img.shape
    cap = int(amd.rt_ppm_max_bytes(w, h))
    buf = np.empty(cap, np.uint8)
    n = C.c_uint64()
    rc = amd.rt_ppm_encode_host(device, _ptr(img), w, h, _ptr(buf), cap, C.byref(n))
    if rc != 0:
        raise RtError(rc, amd.rt_last_error().decode())
    return buf[: n.value].tobytes()


class Canvas:
    """Canvas (src/canvas.rs:22-140): f64 colours, row-major; read_pixel returns None outside the image."""

    def __init__(self, width=0, height=0, _handle=None):
        _, self._host = libraries()
        if _handle is None:
            h = C.c_void_p()
            rc = self._host.rth_canvas_new(width, height, C.byref(h))
            if rc != 0:
                raise RtError(rc, self._host.rth_last_error().decode())
            _handle = h
        self._c = _handle
        w, hh = C.c_uint64(), C.c_uint64()
        self._host.rth_canvas_size(self._c, C.byref(w), C.byref(hh))
        self.width, self.height = int(w.value), int(hh.value)

    def __del__(self):
        if getattr(self, "_c", None):
            self._host.rth_canvas_free(self._c)
            self._c = None

    def write_pixel(self, x, y, rgb):
        self._host.rth_canvas_write_pixel(self._c, x, y, (C.c_double * 3)(*rgb))

    def read_pixel(self, x, y):
        out = (C.c_double * 3)()
        return tuple(out) if self._host.rth_canvas_read_pixel(self._c, x, y, out) else None

    def pixels(self):
        """(H, W, 3) float64 copy of the canvas."""
        if self.width * self.height == 0:
            return np.zeros((self.height, self.width, 3))
        p = self._host.rth_canvas_pixels(self._c)
        return np.ctypeslib.as_array(p, shape=(self.height, self.width, 3)).copy()

    def save(self, path):
        rc = self._host.rth_canvas_save(self._c, str(path).encode())
        if rc != 0:
            raise RtError(rc, self._host.rth_last_error().decode())


class SharedFrameRing:
    """`depth` frames of width x height RGB8 in a POSIX shared-memory file that every process of a job maps
    and page-locks (rt_host_register), so that each GPU's process copies its own bands of the frame over its
    own PCIe link and the finished frame lies in one place.  create=True makes (and at close removes) the file.
    register=False leaves the memory pageable (host-only uses and tests without a GPU)."""

    def __init__(self, name, width, height, depth, create, register=True):
        import mmap
        dev = libraries()[0] if register else None
        self._dev = dev
        self.width, self.height, self.depth = int(width), int(height), int(depth)
        self.frame_bytes = self.width * self.height * 3
        self.bytes = self.frame_bytes * self.depth
        self.path = os.path.join("/dev/shm", name)
        self._owner = bool(create)
        fd = os.open(self.path, (os.O_CREAT | os.O_EXCL | os.O_RDWR) if create else os.O_RDWR, 0o600)
        try:
            if create:
                os.ftruncate(fd, self.bytes)
            elif os.fstat(fd).st_size != self.bytes:
                raise ValueError("shared frame ring %s has another size" % self.path)
            self._map = mmap.mmap(fd, self.bytes)
        finally:
            os.close(fd)
        self._view = (C.c_uint8 * self.bytes).from_buffer(self._map)
        self.address = C.addressof(self._view)
        self._registered = False
        if register and not self.register():
            msg = self._dev.rt_last_error().decode()
            self.close()
            raise RtError(-3, msg)

    def register(self):
        """Page-locks the mapping for DMA from the GPUs of this process; False when the machine refuses."""
        if not self._registered:
            if self._dev is None:
                self._dev = libraries()[0]
            self._registered = self._dev.rt_host_register(C.c_void_p(self.address), self.bytes) == 0
        return self._registered

    def unregister(self):
        if self._registered:
            self._dev.rt_host_unregister(C.c_void_p(self.address))
            self._registered = False

    def frame(self, slot):
        """(H, W, 3) uint8 view of ring slot `slot`."""
        a = np.frombuffer(self._view, dtype=np.uint8, count=self.frame_bytes, offset=slot * self.frame_bytes)
        return a.reshape(self.height, self.width, 3)

    def close(self):
        if getattr(self, "_registered", False):
            self.unregister()
        if getattr(self, "_view", None) is not None:
            self._view = None
            try:
                self._map.close()
            except BufferError:
                pass  # a view handed out by frame() is still alive; the mapping goes with it
            self._map = None
        if getattr(self, "_owner", False):
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            self._owner = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostFrameRing:
    """`depth` frames of width x height RGB8 in page-locked memory of this process (rt_host_alloc): the ring a
    process-per-GPU loop fills with its own bands where the machine cannot page-lock a shared mapping."""

    def __init__(self, width, height, depth):
        dev, _ = libraries()
        self._dev = dev
        self.width, self.height, self.depth = int(width), int(height), int(depth)
        self.frame_bytes = self.width * self.height * 3
        self.bytes = self.frame_bytes * self.depth
        self.address = dev.rt_host_alloc(self.bytes)
        if not self.address:
            raise RtError(-2, "no page-locked host memory for the frame ring")
        C.memset(self.address, 0, self.bytes)

    def frame(self, slot):
        buf = (C.c_uint8 * self.frame_bytes).from_address(self.address + slot * self.frame_bytes)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.height, self.width, 3)

    def close(self):
        if getattr(self, "address", None):
            self._dev.rt_host_free(C.c_void_p(self.address))
            self.address = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FrameLoop:
    """The host library's render loop over several GPUs of this process (host/rt_host.hpp FrameLoop):
    one worker thread and one copy of the scene per part, each GPU copying its own bands of pixel rows
    (band_rows: 8 unless RT_AMD_BAND_ROWS says 16 or 32) into a ring of page-locked host frames shared by all parts.

    With `ring` (a SharedFrameRing or HostFrameRing) the loop renders only parts first_part .. first_part + len(devices) - 1
    of total_parts and assembles the frames in the ring: one process per GPU, no collective."""

    def __init__(self, scene, devices, depth=4, first_part=0, total_parts=None, ring=None):
        _, self._host = libraries()
        self._scene = scene  # keeps the World alive
        self._ring = ring    # and the shared mapping
        self.devices = [int(d) for d in devices]
        self.depth = depth
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        h = C.c_void_p()
        if ring is None and first_part == 0 and total_parts in (None, len(self.devices)):
            rc = self._host.rth_frame_loop_new(scene._h, len(self.devices), arr, depth, C.byref(h))
        else:
            if ring is not None and (ring.depth < depth or (ring.width, ring.height) != (scene.camera().hsize, scene.camera().vsize)):
                raise ValueError("the shared ring does not fit the loop's frames")
            rc = self._host.rth_frame_loop_new_shared(scene._h, len(self.devices), arr, depth, int(first_part),
                                                      int(total_parts or 0), C.c_void_p(ring.address if ring is not None else None),
                                                      C.byref(h))
        if rc != 0:
            raise RtError(rc, self._host.rth_last_error().decode())
        self._l = h
        cam = scene.camera()
        self.width, self.height = cam.hsize, cam.vsize

    def close(self):
        if getattr(self, "_l", None):
            self._host.rth_frame_loop_free(self._l)
            self._l = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, warmup, frames, to_host=True):
        """Returns (seconds of the timed frames, [RtStats per part])."""
        sec = C.c_double()
        stats = (RtStats * len(self.devices))()
        rc = self._host.rth_frame_loop_run(self._l, warmup, frames, 1 if to_host else 0, C.byref(sec), stats)
        if rc != 0:
            raise RtError(rc, self._host.rth_last_error().decode())
        return sec.value, list(stats)

    def blocking_frames(self, frames=1, counters=False):
        """Frames rendered one at a time, nothing overlapping; [RtStats per part], summed over the frames."""
        stats = (RtStats * len(self.devices))()
        rc = self._host.rth_frame_loop_blocking(self._l, frames, 1 if counters else 0, stats)
        if rc != 0:
            raise RtError(rc, self._host.rth_last_error().decode())
        return list(stats)

    def host_frame(self, slot):
        """(H, W, 3) uint8 view of ring slot `slot` (page-locked memory owned by the loop)."""
        p = self._host.rth_frame_loop_host_frame(self._l, slot)
        buf = (C.c_uint8 * (self.width * self.height * 3)).from_address(p)
        return np.frombuffer(buf, dtype=np.uint8).reshape(self.height, self.width, 3)

    @property
    def band_rows(self):
        """Rows of a band of this loop's parts (8, or what RT_AMD_BAND_ROWS asked for when the loop was made)."""
        self._host.rth_frame_loop_band_rows.restype = C.c_uint64
        self._host.rth_frame_loop_band_rows.argtypes = [C.c_void_p]
        return int(self._host.rth_frame_loop_band_rows(self._l))

    @property
    def frames_per_launch(self):
        """Frames of a part the last run() put into one kernel launch (rt_render_frames): 1 for a whole frame and for
        scenes that spawn rays, 2 for halves, 4 from four parts on, at most half the ring; RT_AMD_FRAMES_PER_LAUNCH."""
        self._host.rth_frame_loop_frames_per_launch.restype = C.c_uint64
        self._host.rth_frame_loop_frames_per_launch.argtypes = [C.c_void_p]
        return int(self._host.rth_frame_loop_frames_per_launch(self._l))

    def info(self):
        out = (C.c_uint64 * 2)()
        b = C.c_double()
        self._host.rth_frame_loop_info(self._l, out, C.byref(b))
        return {"scene_bytes": int(out[0]), "frame_bytes": int(out[1]), "build_seconds": b.value}


class Scene:
    """A World and a Camera loaded from the reference's scene JSON, resident on one GPU.

    device=None parses and flattens only (host logic; needs no GPU).
    """

    def __init__(self, text, base_dir="", device=0):
        self._amd, self._host = libraries()
        h = C.c_void_p()
        rc = self._host.rth_scene_from_json(text.encode(), base_dir.encode(), -1 if device is None else device, C.byref(h))
        if rc != 0:
            raise RtError(rc, self._host.rth_last_error().decode())
        self._h = h
        self.device = device

    @classmethod
    def from_file(cls, path, base_dir=None, device=0):
        with open(path) as f:
            text = f.read()
        return cls(text, os.getcwd() if base_dir is None else base_dir, device)

    def close(self):
        if getattr(self, "_h", None):
            self._host.rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- description
    def info(self):
        out = (C.c_uint64 * 8)()
        self._host.rth_scene_info(self._h, out)
        keys = ("width", "height", "leaves", "gates", "xforms", "triangles", "materials", "patterns")
        return dict(zip(keys, [int(v) for v in out]))

    @property
    def width(self):
        return self.info()["width"]

    @property
    def height(self):
        return self.info()["height"]

    def camera(self):
        cam = getattr(self, "_camera", None)
        if cam is None:
            cam = RtCamera()
            rc = self._host.rth_scene_camera(self._h, C.byref(cam))
            if rc != 0:
                raise RtError(rc, self._host.rth_last_error().decode())
            self._camera = cam
        return cam

    def set_canvas(self, width, height):
        self._camera = None
        self._host.rth_scene_set_canvas(self._h, width, height)

    def set_camera(self, width, height, fov, cam_from, cam_to, cam_up):
        f = (C.c_double * 3)(*cam_from[:3])
        t = (C.c_double * 3)(*cam_to[:3])
        up = list(cam_up) + [0.0] * (4 - len(cam_up))
        u = (C.c_double * 4)(*up[:4])
        self._camera = None
        self._host.rth_scene_set_camera(self._h, width, height, fov, f, t, u)

    def set_transform(self, path, matrix16):
        """Shape::set_transform (src/shape.rs:592-656) on the shape at `path`: path[0] indexes the world's
        objects, further entries the children of groups / CSG operators.  Takes effect with update()."""
        p = (C.c_uint64 * len(path))(*path)
        m = (C.c_double * 16)(*[float(v) for v in matrix16])
        rc = self._host.rth_scene_set_transform(self._h, p, len(path), m)
        if rc != 0:
            raise RtError(rc, self._host.rth_last_error().decode())

    def update(self):
        """Moves the changed world onto the device: rt_scene_update, a refit of the hierarchy (no rebuild)."""
        rc = self._host.rth_scene_update(self._h)
        if rc != 0:
            raise RtError(rc, self._host.rth_last_error().decode())

    def footprint(self):
        b, s = C.c_uint64(), C.c_double()
        self._check(self._amd.rt_scene_footprint(self._dev(), C.byref(b), C.byref(s)))
        n, t = C.c_uint64(), C.c_double()
        self._check(self._amd.rt_scene_build_info(self._dev(), C.byref(n), C.byref(t)))
        return {"scene_bytes": b.value, "build_seconds": s.value, "device_builds": n.value, "device_build_seconds": t.value}

    def _dev(self):
        d = self._host.rth_scene_device(self._h)
        if not d:
            raise RtError(-3, "scene was loaded without a device")
        return d

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self._amd.rt_last_error().decode())

    # ---- rendering
    def render(self, part=0, n_parts=1, counters=False, rgb8=True, rgb64=True, hit_leaf=True):
        """parallel_render without the file: returns (dict of host arrays, RtStats)."""
        cam = self.camera()
        w, h = cam.hsize, cam.vsize
        out = {}
        a8 = np.zeros((h, w, 3), np.uint8) if rgb8 else None
        a64 = np.zeros((h, w, 3), np.float64) if rgb64 else None
        al = np.full((h, w), -1, np.int32) if hit_leaf else None
        st = RtStats()
        self._check(self._amd.rt_render_host(self._dev(), C.byref(cam), part, n_parts, RT_FLAG_COUNTERS if counters else 0,
                                             _ptr(a8), _ptr(a64), _ptr(al), C.byref(st)))
        if rgb8:
            out["rgb8"] = a8
        if rgb64:
            out["rgb64"] = a64
        if hit_leaf:
            out["hit_leaf"] = al
        return out, st

    def render_device(self, rgb8_ptr=0, rgb64_ptr=0, hit_leaf_ptr=0, part=0, n_parts=1, counters=False, wait=True,
                      compact=False, bands=False, band_rows=32):
        """rt_render into device buffers given by address (e.g. torch tensor .data_ptr()).  With
        wait=False the frame is only enqueued (RT_FLAG_ASYNC); collect with wait()."""
        cam = self.camera()
        st = RtStats()
        flags = ((RT_FLAG_COUNTERS if counters else 0) | (0 if wait else RT_FLAG_ASYNC) | (RT_FLAG_COMPACT if compact else 0)
                 | (RT_FLAG_BANDS if bands else 0) | ({32: 0, 16: RT_FLAG_BANDS_16, 8: RT_FLAG_BANDS_8}[band_rows] if bands else 0))
        self._check(self._amd.rt_render(self._dev(), C.byref(cam), part, n_parts, flags,
                                        C.c_void_p(rgb8_ptr or None), C.c_void_p(rgb64_ptr or None),
                                        C.c_void_p(hit_leaf_ptr or None), C.byref(st)))
        return st

    def render_frames_device(self, cameras, rgb8_ptrs, part=0, n_parts=1, bands=False, band_rows=32):
        """rt_render_frames: frame k through cameras[k] (RtCamera structs, e.g. copies of camera() taken between
        set_camera calls) into the device buffer at rgb8_ptrs[k]; enqueued, collect with wait()."""
        n = len(cameras)
        cams = (RtCamera * n)(*cameras)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(p) for p in rgb8_ptrs])
        flags = (RT_FLAG_ASYNC | (RT_FLAG_BANDS if bands else 0)
                 | ({32: 0, 16: RT_FLAG_BANDS_16, 8: RT_FLAG_BANDS_8}[band_rows] if bands else 0))
        self._check(self._amd.rt_render_frames(self._dev(), cams, n, part, n_parts, flags, ptrs))

    def wait(self):
        """rt_scene_wait: blocks until the enqueued frames are complete; sums of their statistics."""
        st = RtStats()
        self._check(self._amd.rt_scene_wait(self._dev(), C.byref(st)))
        return st

    def download(self, src_ptr, dst_ptr, nbytes, after_stream=0):
        """rt_download: device -> pinned host copy, enqueued on the scene's copy stream after the work
        now on after_stream (address of a hipStream_t; 0 = the scene's render stream)."""
        self._check(self._amd.rt_download(self._dev(), C.c_void_p(src_ptr), C.c_void_p(dst_ptr), nbytes,
                                          C.c_void_p(after_stream or None)))

    def download_2d(self, src_ptr, dst_ptr, row_bytes, rows, pitch, after_stream=0):
        """rt_download_2d: `rows` pieces of row_bytes bytes, pitch bytes apart in both buffers."""
        self._check(self._amd.rt_download_2d(self._dev(), C.c_void_p(src_ptr), C.c_void_p(dst_ptr), row_bytes, rows, pitch,
                                             C.c_void_p(after_stream or None)))

    def copy_stream_ptr(self):
        p = C.c_void_p()
        self._check(self._amd.rt_scene_copy_stream(self._dev(), C.byref(p)))
        return p.value or 0

    def stream_ptr(self):
        """Address of the hipStream_t the scene's work runs on (for torch.cuda.ExternalStream)."""
        p = C.c_void_p()
        self._check(self._amd.rt_scene_stream(self._dev(), C.byref(p)))
        return p.value or 0

    def parallel_render(self, jobs, out_path):
        rc = self._host.rth_parallel_render(self._h, jobs, str(out_path).encode())
        if rc != 0:
            raise RtError(rc, self._host.rth_last_error().decode())

    def render_canvas(self, jobs=1):
        """The canvas parallel_render holds before it saves (src/parallel.rs:185-200)."""
        h = C.c_void_p()
        rc = self._host.rth_render_canvas(self._h, jobs, C.byref(h))
        if rc != 0:
            raise RtError(rc, self._host.rth_last_error().decode())
        return Canvas(_handle=h)

    def frame_loop(self, devices, depth=4, first_part=0, total_parts=None, ring=None):
        """parallel_render's workers kept alive over frames: part first_part + k of total_parts (default
        len(devices)) on GPU devices[k]; frames assembled in `ring` (a SharedFrameRing) when given."""
        return FrameLoop(self, devices, depth, first_part, total_parts, ring)

    # ---- batch forms of the reference's per-ray API
    def ray_for_pixel(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.uint32).reshape(-1, 2)
        rays = np.zeros((len(xy), 6), np.float64)
        cam = self.camera()
        self._check(self._amd.rt_ray_for_pixel(self._dev(), C.byref(cam), len(xy), _ptr(xy), _ptr(rays)))
        return rays

    def color_at(self, rays, remaining=5, counters=False):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        rgb = np.zeros((len(rays), 3), np.float64)
        leaf = np.full(len(rays), -1, np.int32)
        st = RtStats()
        self._check(self._amd.rt_color_at(self._dev(), len(rays), _ptr(rays), remaining,
                                          RT_FLAG_COUNTERS if counters else 0, _ptr(rgb), _ptr(leaf), C.byref(st)))
        return rgb, leaf, st

    def intersect_hit(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        t = np.zeros(len(rays), np.float64)
        leaf = np.full(len(rays), -1, np.int32)
        self._check(self._amd.rt_intersect_hit(self._dev(), len(rays), _ptr(rays), _ptr(t), _ptr(leaf)))
        return t, leaf

    def is_shadowed(self, points):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(len(pts), np.uint8)
        self._check(self._amd.rt_is_shadowed(self._dev(), len(pts), _ptr(pts), _ptr(out)))
        return out.astype(bool)
