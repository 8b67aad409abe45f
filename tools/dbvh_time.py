import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import scenegen, tempfile
from __graft_entry__ import load_package
pkg = load_package()
d = tempfile.mkdtemp()
path, tris = scenegen.heightfield(d, cells=708, name="hf")
text = open(path).read()
os.environ["RT_AMD_DEVICE_BVH"] = "1"; os.environ["RT_AMD_BUILD_TIMING"] = "1"
for k in range(2):
    t0 = time.perf_counter(); s = pkg.Scene(text, '/root/repo', device=0); t1 = time.perf_counter()
    f = s.footprint(); print("create %.1f ms, build phase %.1f ms, device %.2f ms" % ((t1 - t0) * 1e3, f["build_seconds"] * 1e3, f["device_build_seconds"] * 1e3), flush=True)
    st = s.render_device(rgb8_ptr=0, hit_leaf_ptr=0) if False else None
os.environ["RT_AMD_DEVICE_BVH"] = "0"
t0 = time.perf_counter(); s = pkg.Scene(text, '/root/repo', device=0); t1 = time.perf_counter()
f = s.footprint(); print("host: create %.1f ms, build phase %.1f ms" % ((t1 - t0) * 1e3, f["build_seconds"] * 1e3))
