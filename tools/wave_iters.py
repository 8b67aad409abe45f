#!/usr/bin/env python3
"""Joins the two diagnostic frames of one scene: per-wave run time (librt_amd_wclk.so) and per-wave loop
iterations (librt_amd_wit.so: node visits | triangle tests << 12), both one value per 8x8 pixel block.
Usage: wave_iters.py clock.npy iters.npy"""
import sys

import numpy as np

clock = np.load(sys.argv[1]).astype(np.float64)
it = np.load(sys.argv[2]).astype(np.int64)
nodes, tris = it & 4095, it >> 12
m = clock > 0
c, n, t = clock[m], nodes[m], tris[m]
print("waves %d; ticks mean %.0f max %.0f; node iterations mean %.1f max %d; triangle iterations mean %.1f max %d" % (
    c.size, c.mean(), c.max(), n.mean(), n.max(), t.mean(), t.max()))
A = np.stack([np.ones_like(c), n, t], axis=1)
coef, *_ = np.linalg.lstsq(A, c, rcond=None)
print("least squares: ticks = %.0f + %.0f per node iteration + %.0f per triangle iteration (rms residual %.0f)" % (
    coef[0], coef[1], coef[2], np.sqrt(np.mean((A @ coef - c) ** 2))))
order = np.argsort(c)[::-1]
for k in order[:8]:
    print("  wave with %7.0f ticks: %4d node iterations, %4d triangle iterations" % (c[k], n[k], t[k]))
for lo, hi in [(0, 1), (1, 8), (8, 32), (32, 128), (128, 100000)]:
    sel = (n + t >= lo) & (n + t < hi)
    if sel.any():
        print("  waves with %d..%d iterations: %d, mean %.0f ticks, share of all wave time %.3f" % (
            lo, hi - 1, sel.sum(), c[sel].mean(), c[sel].sum() / c.sum()))
