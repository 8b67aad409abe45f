#!/usr/bin/env python3
"""Probe: the render kernel writing its pixels straight into page-locked host memory (no device frame,
no copy) against render into HBM + DMA copy.  Teapot 1080p, frames enqueued back to back."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
scene = pkg.Scene.from_file(os.path.join(ROOT, "scenes", "teapot.json"), base_dir=ROOT, device=0)
W, H = 1920, 1080
scene.set_canvas(W, H)
dev = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0") for _ in range(4)]
host = [torch.zeros((H, W, 3), dtype=torch.uint8).pin_memory() for _ in range(4)]
want = scene.render(rgb64=False, hit_leaf=False)[0]["rgb8"]


def loop(n, direct):
    for k in range(n):
        if direct:
            scene.render_device(rgb8_ptr=host[k % 4].data_ptr(), wait=False)
        else:
            scene.render_device(rgb8_ptr=dev[k % 4].data_ptr(), wait=False)
            scene.download(dev[k % 4].data_ptr(), host[k % 4].data_ptr(), W * H * 3)
    scene.wait()
    torch.cuda.synchronize()


for direct in (False, True, False, True):
    loop(30, direct)
    t0 = time.perf_counter()
    loop(100, direct)
    dt = time.perf_counter() - t0
    ok = bool((host[0].numpy() == want).all())
    print("%-28s %.4f ms/frame  %.1f GB/s to the host  frame correct: %s" % (
        "kernel stores to host memory" if direct else "HBM frame + DMA copy", dt / 100 * 1e3, W * H * 3 * 100 / dt / 1e9, ok), flush=True)
