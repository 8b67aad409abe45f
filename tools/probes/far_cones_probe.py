#!/usr/bin/env python3
"""Pixels that differ from the oracle when the camera is 1e2 .. 1e7 units from a crowd that holds bounded cones
(the widening of the boxes by the reference's rounding is off for such scenes: their parallel-ray branch puts
hits far out on the cones' extensions).  Prints one line per distance."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from test_gpu_edges import far_cone_crowd, FAR_CAMERAS  # noqa: E402

pkg = load_package()
for dist, fov in FAR_CAMERAS:
    base = json.load(open(os.path.join(ROOT, "scenes", "example.json")))
    base.update(canvas_width=96, canvas_height=64, shapes=far_cone_crowd(), camera_from=[0.0, 0.3 * dist, -dist],
                camera_to=[0.0, 0.0, 0.0], field_of_view=fov,
                light={"position": [-0.5 * dist, dist, -0.5 * dist], "intensity": [1.0, 1.0, 1.0]})
    text = json.dumps(base)
    out, st = pkg.Scene(text, ROOT, device=0).render()
    want = orc.OracleScene(text, ROOT).render(jobs=8)
    leaf = int((out["hit_leaf"] != want["hit_leaf"]).sum())
    mask = int(((out["hit_leaf"] >= 0) != (want["hit_leaf"] >= 0)).sum())
    px = int((out["rgb8"] != orc.quantise(want["rgb64"])).any(axis=2).sum())
    print("distance %.0e: pixels whose hit leaf differs %d (hit / miss differs %d), whose 8-bit colour differs %d, hits %d" % (
        dist, leaf, mask, px, int((want["hit_leaf"] >= 0).sum())), flush=True)
