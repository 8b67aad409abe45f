#!/usr/bin/env python3
"""Times many short stretches of the frame loop in one process and lists the slow ones with the time they
began at: does the loop stall now and then, how long, and is it periodic?  usage: stall_probe.py [stretches] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
to_host = os.environ.get("STALL_TO_HOST", "1") != "0"
scene = pkg.Scene.from_file(os.path.join(ROOT, "scenes", "teapot.json"), base_dir=ROOT, device=None)
scene.set_canvas(1920, 1080)
loop = scene.frame_loop([0], depth=4)
if os.environ.get("STALL_PRIME", "1") != "0":
    loop.run(64, 20, to_host)
t_begin = time.perf_counter()
rows = []
for k in range(n):
    t0 = time.perf_counter() - t_begin
    s, _ = loop.run(int(os.environ.get("STALL_WARM", "5")), frames, to_host)
    rows.append((t0, s * 1e3))
ms = sorted(r[1] for r in rows)
med = ms[len(ms) // 2]
print("stretches %d of %d frames, to_host %s: median %.3f ms, min %.3f, max %.3f" % (n, frames, to_host, med, ms[0], ms[-1]))
for t0, m in rows:
    if m > 1.15 * med:
        print("  slow stretch %d at %.3f s: %.3f ms (+%.3f)" % (rows.index((t0, m)), t0, m, m - med))
loop.close()
