"""MI355X-native render path (camera ray -> intersection -> Phong shading with
shadow, reflection and refraction rays) behind the reference ray tracer's
Camera / World / parallel_render interface.

The work is done by two in-tree shared libraries built by ``make`` in this
directory: ``lib/librt_amd.so`` (HIP kernels + the C ABI of include/rt_amd.h)
and ``lib/librt_host.so`` (scene JSON, OBJ, flattening, PPM).  This package is
a thin ctypes binding; there is no CPU fallback: a missing library raises.
"""
from . import bindings, tiles  # noqa: F401
from .bindings import (  # noqa: F401
    Canvas,
    RtCamera,
    RtStats,
    RtError,
    Scene,
    device_count,
    libraries,
    ppm_encode,
)
