#!/usr/bin/env python3
"""Debugging aid: follow one pixel's ray tree through the CPU oracle and replay every ray of it on
the GPU path, printing where the two first disagree.

  python tools/trace_pixel.py scenes/glass.json X Y [--size W H]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
import oracle as orc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("x", type=int)
    ap.add_argument("y", type=int)
    ap.add_argument("--size", type=int, nargs=2)
    args = ap.parse_args()
    pkg = entry.load_package()
    gpu = pkg.Scene.from_file(args.scene, base_dir=ROOT, device=0)
    ref = orc.OracleScene.from_file(args.scene, base_dir=ROOT)
    if args.size:
        gpu.set_canvas(*args.size)
        ref.set_canvas(*args.size)
    ray = ref.ray_for_pixel(np.array([[args.x, args.y]], np.uint32))[0]
    np.set_printoptions(linewidth=200, precision=12)
    for rec in ref.trace(ray, 5):
        one = rec[:6].reshape(1, 6)
        rem = int(rec[6])
        rgb, leaf, _ = gpu.color_at(one, remaining=rem)
        surf, _, _ = gpu.color_at(one, remaining=0)
        want_surf, _, _ = ref.color_at(one, remaining=0)
        t, _ = gpu.intersect_hit(one)
        print("remaining %d  leaf %d / %d  t %.17g / %.17g  n1 %.6g n2 %.6g  |colour diff| %.3g  |surface diff| %.3g"
              % (rem, int(rec[7]), leaf[0], rec[8], t[0], rec[9], rec[10], np.abs(rgb[0] - rec[11:14]).max(),
                 np.abs(surf - want_surf).max()))


if __name__ == "__main__":
    main()
