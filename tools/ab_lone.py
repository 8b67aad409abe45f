#!/usr/bin/env python3
"""Lone launches of the teapot frame and of one part of eight, and the pipelined loop with the frames left in HBM:
what a change does to the latency of a launch as against its throughput.  RT_AMD_LIB / RT_HOST_LIB select the build."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
scene = pkg.Scene.from_file(os.path.join(ROOT, "scenes", "teapot.json"), base_dir=ROOT, device=None)
scene.set_canvas(1920, 1080)
row = {"lib": os.environ.get("RT_AMD_LIB", "default")}
for parts, part in ((1, 0), (8, 0), (8, 3)):
    ring = pkg.bindings.HostFrameRing(1920, 1080, 4)
    loop = scene.frame_loop([0], depth=4, first_part=part, total_parts=parts, ring=ring)
    loop.run(0, 200, False)
    sec, _ = loop.run(0, 400, False)
    lone = loop.blocking_frames(40, False)[0]
    row["part %d of %d" % (part, parts)] = {"hbm_loop_ms": round(sec / 400 * 1e3, 4), "lone_kernel_ms": round(lone.kernel_ms / 40, 4)}
    loop.close()
    ring.close()
print(json.dumps(row), flush=True)
