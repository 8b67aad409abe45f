#!/usr/bin/env python3
"""Debugging aid: renders random scenes of tests/test_gpu_fuzz.py on both paths, lists the pixels that
differ and replays every ray of those pixels' ray trees on the GPU path.

  python tools/fuzz_debug.py SEED [SEED ...]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
import oracle as orc  # noqa: E402
import test_gpu_fuzz as fuzz  # noqa: E402


def main():
    pkg = entry.load_package()
    np.set_printoptions(linewidth=200, precision=17)
    for seed in [int(a) for a in sys.argv[1:]]:
        text = fuzz.random_scene(seed)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "fuzz_%d.json" % seed), "w").write(text)
        gpu = pkg.Scene(text, base_dir=ROOT, device=0)
        ref = orc.OracleScene(text, base_dir=ROOT)
        out, _ = gpu.render()
        want = ref.render(jobs=8)
        d = np.abs(out["rgb64"] - want["rgb64"]).max(axis=2)
        ys, xs = np.nonzero(d > 1e-10)
        print("seed %d: %d pixels differ" % (seed, len(ys)))
        for y, x in list(zip(ys, xs))[:4]:
            print(" pixel (%d, %d): gpu %s  cpu %s" % (x, y, out["rgb64"][y, x], want["rgb64"][y, x]))
            ray = ref.ray_for_pixel(np.array([[x, y]], np.uint32))[0]
            for rec in ref.trace(ray, 5):
                one = rec[:6].reshape(1, 6)
                rem = int(rec[6])
                rgb, leaf, _ = gpu.color_at(one, remaining=rem)
                surf, _, _ = gpu.color_at(one, remaining=0)
                want_surf, _, _ = ref.color_at(one, remaining=0)
                t, _ = gpu.intersect_hit(one)
                print("  remaining %d  leaf %d / %d  t %.17g / %.17g  n1 %.6g n2 %.6g  |colour diff| %.3g  |surface diff| %.3g"
                      % (rem, int(rec[7]), leaf[0], rec[8], t[0], rec[9], rec[10], np.abs(rgb[0] - rec[11:14]).max(),
                         np.abs(surf - want_surf).max()))
                print("     ray o %s d %s" % (rec[:3], rec[3:6]))


if __name__ == "__main__":
    main()
