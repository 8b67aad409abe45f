#!/usr/bin/env python3
"""Lone (blocking) launches of the whole teapot frame and of part 0 of 8: mean kernel time over 50 launches each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
scene = pkg.Scene.from_file(os.path.join(ROOT, "scenes", "teapot.json"), base_dir=ROOT, device=None)
scene.set_canvas(1920, 1080)
out = []
for n in (1, 2, 8):
    ring = pkg.bindings.HostFrameRing(1920, 1080, 4)
    loop = scene.frame_loop([0], depth=4, first_part=0, total_parts=n, ring=ring)
    loop.run(0, 200, False)
    lone = loop.blocking_frames(50, False)[0]
    out.append("part 0 of %d: %.4f ms" % (n, lone.kernel_ms / 50))
    loop.close()
    ring.close()
print("; ".join(out), flush=True)
