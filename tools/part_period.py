#!/usr/bin/env python3
"""Frame period of ONE GPU's share of the frame when the frame is split into N parts (bands of 32, 16 or 8 rows):
every part of N rendered alone on GPU 0 with its copies to the host, as each GPU of an N-GPU run does.  What an
N-GPU node can reach at best is the slowest part's period; this needs one GPU only.  Per N and band height: the
slowest part, and how unevenly the rays fall on the parts (max / min).
usage: part_period.py [depth] [band rows ...]   (depth 0 or none: what bench.py takes for a long run, 16)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

DEPTH_ARG = int(sys.argv[1]) if len(sys.argv) > 1 else 0
HEIGHTS = [int(v) for v in sys.argv[2:]] or [32, 16, 8]
pkg = load_package()
scene = pkg.Scene.from_file(os.path.join(ROOT, "scenes", "teapot.json"), base_dir=ROOT, device=None)
scene.set_canvas(1920, 1080)
for rows in HEIGHTS:
    os.environ["RT_AMD_BAND_ROWS"] = str(rows)  # read when a loop is made
    for n in (1, 2, 4, 8):
        if n == 1 and rows != HEIGHTS[0]:
            continue  # one part is the whole frame whatever the band height
        worst, rays = None, []
        DEPTH = DEPTH_ARG or 16
        for part in range(n):
            loop = scene.frame_loop([0], depth=DEPTH, first_part=part, total_parts=n)  # the loop's own ring, as in bench.py
            assert loop.band_rows == rows
            loop.run(0, 100, True)
            sec, st = loop.run(0, 400, True)
            sec_hbm, _ = loop.run(0, 400, False)
            loop.run(0, 100, False)
            lone = loop.blocking_frames(40, False)[0]
            row = {"band_rows": rows, "parts": n, "part": part, "frames_per_launch": loop.frames_per_launch, "ms_per_frame_host": sec / 400 * 1e3, "ms_per_frame_hbm": sec_hbm / 400 * 1e3,
                   "lone_kernel_ms": lone.kernel_ms / 40, "rays": (st[0].primary + st[0].shadow + st[0].reflect + st[0].refract) / 400}
            print(json.dumps(row), flush=True)
            rays.append(row["rays"])
            worst = row if worst is None or row["ms_per_frame_host"] > worst["ms_per_frame_host"] else worst
            loop.close()
        print("# depth %d, bands of %d rows, %d parts: slowest part %.4f ms per frame on the host, %.4f left in HBM, lone launch %.4f; "
              "rays per part max / min %.3f" % (DEPTH, rows, n, worst["ms_per_frame_host"], worst["ms_per_frame_hbm"],
                                                 worst["lone_kernel_ms"], max(rays) / min(rays)), flush=True)
