"""The cone direction index built by the GPU kernels equals the one the host builder makes."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_bands,seed", [(10, 1), (33, 2), (200, 3), (2500, 4), (25000, 5)])
def test_device_index_equals_host_index(pkg, n_bands, seed):
    amd, _ = pkg.libraries()
    fn = amd.rt_selftest_band_grid_device
    fn.argtypes = [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_double),
                   C.POINTER(C.c_double)]
    n, entries, host_s, dev_s = C.c_int(), C.c_uint64(), C.c_double(), C.c_double()
    rc = fn(n_bands, seed, 0, C.byref(n), C.byref(entries), C.byref(host_s), C.byref(dev_s))
    print("bands %d: grid %d, %d entries, host %.3f s, device %.3f s" % (n_bands, n.value, entries.value, host_s.value, dev_s.value))
    assert rc == 0, amd.rt_last_error()
    assert (n.value == 0) == (n_bands <= 32)  # a short list is scanned, not indexed


def test_scene_renders_the_same_with_either_builder(pkg, root, monkeypatch):
    """A crowd of cones rendered from a scene whose index was built on the device and from one built on
    the host: identical frames."""
    import numpy as np
    import test_gpu_fuzz as fuzz
    text = fuzz.crowd_scene(7)
    a, _ = pkg.Scene(text, base_dir=root, device=0).render()
    monkeypatch.setenv("RT_AMD_BAND_GRID_HOST", "1")
    b, _ = pkg.Scene(text, base_dir=root, device=0).render()
    assert np.array_equal(a["rgb64"], b["rgb64"], equal_nan=True) and np.array_equal(a["hit_leaf"], b["hit_leaf"])
