#!/usr/bin/env python3
"""Frame period of ONE GPU's share of the frame when the frame is split into N parts (bands of 32 rows):
part 0 of N rendered alone on GPU 0 with its copies to the host, as each GPU of an N-GPU run does.  What an
N-GPU node can reach at best is the slowest part's period; this needs one GPU only."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

DEPTH = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pkg = load_package()
scene = pkg.Scene.from_file(os.path.join(ROOT, "scenes", "teapot.json"), base_dir=ROOT, device=None)
scene.set_canvas(1920, 1080)
for n in (1, 2, 4, 8):
    worst = None
    for part in range(n):
        ring = pkg.bindings.HostFrameRing(1920, 1080, DEPTH)
        loop = scene.frame_loop([0], depth=DEPTH, first_part=part, total_parts=n, ring=ring)
        loop.run(0, 100, True)
        sec, st = loop.run(0, 400, True)
        sec_hbm, _ = loop.run(0, 400, False)
        lone = loop.blocking_frames(10, False)[0]
        row = {"parts": n, "part": part, "ms_per_frame_host": sec / 400 * 1e3, "ms_per_frame_hbm": sec_hbm / 400 * 1e3,
               "lone_kernel_ms": lone.kernel_ms / 10, "rays": (st[0].primary + st[0].shadow + st[0].reflect + st[0].refract) / 400}
        print(json.dumps(row), flush=True)
        worst = row if worst is None or row["ms_per_frame_host"] > worst["ms_per_frame_host"] else worst
        loop.close()
        ring.close()
    print("# depth %d, %d parts: slowest part %.4f ms per frame on the host, %.4f left in HBM" % (
        DEPTH, n, worst["ms_per_frame_host"], worst["ms_per_frame_hbm"]), flush=True)
