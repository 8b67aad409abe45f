"""ctypes wrapper over the CPU oracle (oracle/_build/liboracle.so).

Test infrastructure only: the oracle restates the reference algorithm on the
CPU and is what GPU results are compared with.  Nothing in the product imports
this module.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            raise RuntimeError("%s missing: run `make -C oracle`" % LIB)
        L = C.CDLL(LIB)
        vp, u64 = C.c_void_p, C.c_uint64
        L.orc_scene_from_json.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        L.orc_scene_from_json.restype = vp
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_info.argtypes = [vp, C.POINTER(u64)]
        L.orc_scene_set_canvas.argtypes = [vp, u64, u64]
        L.orc_scene_set_camera.argtypes = [vp, u64, u64, C.c_double, vp, vp, vp]
        L.orc_render.argtypes = [vp, C.c_int, u64, u64, u64, u64, vp, vp, vp, C.c_char_p, C.c_int]
        L.orc_color_at.argtypes = [vp, u64, vp, C.c_int, vp, vp, vp]
        L.orc_intersect_hit.argtypes = [vp, u64, vp, vp, vp]
        L.orc_trace.argtypes = [vp, vp, C.c_int, vp, u64]
        L.orc_trace.restype = u64
        L.orc_is_shadowed.argtypes = [vp, u64, vp, vp]
        L.orc_ray_for_pixel.argtypes = [vp, u64, vp, vp]
        L.orc_quantise.argtypes = [u64, vp, vp]
        L.orc_ppm_encode.argtypes = [u64, u64, vp, vp, u64]
        L.orc_ppm_encode.restype = u64
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def quantise(rgb64):
    a = np.ascontiguousarray(rgb64, dtype=np.float64)
    out = np.zeros(a.shape, np.uint8)
    lib().orc_quantise(a.size, _p(a), _p(out))
    return out


def ppm_encode(rgb64):
    a = np.ascontiguousarray(rgb64, dtype=np.float64)
    h, w, _ = a.shape
    n = lib().orc_ppm_encode(w, h, _p(a), None, 0)
    buf = C.create_string_buffer(n)
    lib().orc_ppm_encode(w, h, _p(a), buf, n)
    return buf.raw


class OracleError(RuntimeError):
    pass


class OracleScene:
    def __init__(self, text, base_dir=""):
        err = C.create_string_buffer(512)
        self._h = lib().orc_scene_from_json(text.encode(), base_dir.encode(), err, 512)
        if not self._h:
            raise OracleError(err.value.decode())

    @classmethod
    def from_file(cls, path, base_dir=None):
        with open(path) as f:
            return cls(f.read(), os.getcwd() if base_dir is None else base_dir)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_free(self._h)
            self._h = None

    def info(self):
        out = (C.c_uint64 * 4)()
        lib().orc_scene_info(self._h, out)
        return {"width": int(out[0]), "height": int(out[1]), "leaves": int(out[2]), "objects": int(out[3])}

    def set_canvas(self, w, h):
        lib().orc_scene_set_canvas(self._h, w, h)

    def set_camera(self, w, h, fov, cam_from, cam_to, cam_up):
        f = np.array(cam_from[:3], np.float64)
        t = np.array(cam_to[:3], np.float64)
        u = np.array((list(cam_up) + [0.0] * 4)[:4], np.float64)
        lib().orc_scene_set_camera(self._h, w, h, fov, _p(f), _p(t), _p(u))

    def render(self, jobs=1, x0=0, xstep=1, y0=0, ystep=1):
        i = self.info()
        w, h = i["width"], i["height"]
        rgb = np.zeros((h, w, 3), np.float64)
        leaf = np.full((h, w), -1, np.int32)
        counts = (C.c_uint64 * 4)()
        err = C.create_string_buffer(512)
        rc = lib().orc_render(self._h, jobs, x0, xstep, y0, ystep, _p(rgb), _p(leaf), counts, err, 512)
        if rc != 0:
            raise OracleError(err.value.decode())
        return {"rgb64": rgb, "hit_leaf": leaf, "counts": [int(c) for c in counts]}

    def color_at(self, rays, remaining=5):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        rgb = np.zeros((len(rays), 3), np.float64)
        leaf = np.full(len(rays), -1, np.int32)
        counts = (C.c_uint64 * 4)()
        lib().orc_color_at(self._h, len(rays), _p(rays), remaining, _p(rgb), _p(leaf), counts)
        return rgb, leaf, [int(c) for c in counts]

    def trace(self, ray, remaining=5):
        """Every color_at call one ray causes that hit something: rows of
        ray(6), remaining, leaf, t, n1, n2, colour(3) in call order."""
        ray = np.ascontiguousarray(ray, np.float64).reshape(6)
        out = np.zeros((256, 14), np.float64)
        n = lib().orc_trace(self._h, _p(ray), remaining, _p(out), len(out))
        return out[:n]

    def intersect_hit(self, rays):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        t = np.zeros(len(rays), np.float64)
        leaf = np.full(len(rays), -1, np.int32)
        lib().orc_intersect_hit(self._h, len(rays), _p(rays), _p(t), _p(leaf))
        return t, leaf

    def is_shadowed(self, points):
        pts = np.ascontiguousarray(points, np.float64).reshape(-1, 3)
        out = np.zeros(len(pts), np.uint8)
        lib().orc_is_shadowed(self._h, len(pts), _p(pts), _p(out))
        return out.astype(bool)

    def ray_for_pixel(self, xy):
        xy = np.ascontiguousarray(xy, np.uint32).reshape(-1, 2)
        rays = np.zeros((len(xy), 6), np.float64)
        lib().orc_ray_for_pixel(self._h, len(xy), _p(xy), _p(rays))
        return rays
