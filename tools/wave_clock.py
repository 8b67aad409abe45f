#!/usr/bin/env python3
"""Per-wave run times of the fused kernel, read from a frame rendered by the diagnostic build
(make -C ray-tracer_amd diag; -DRT_WAVE_CLOCK_IMAGE=1: every pixel carries its wave's s_memtime ticks in its
three bytes; =2: the wave's loop iterations).  The tick is a shader clock cycle on this GPU (the longest wave of a
lone 0.17 ms launch counts 250 000).  Run with RT_AMD_LIB=librt_amd_wclk.so RT_HOST_LIB=librt_host_wclk.so (or _wit)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pkg = load_package()
name = sys.argv[1] if len(sys.argv) > 1 else "teapot.json"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
scene = pkg.Scene.from_file(os.path.join(ROOT, "scenes", name), base_dir=ROOT, device=0)
scene.set_canvas(W, H)
for parts in (1, 8):
    for rep in range(3):
        out, st = scene.render(part=parts - 1, n_parts=parts, rgb64=False, hit_leaf=False)
    img = out["rgb8"].astype(np.int64)
    ticks = img[..., 0] | (img[..., 1] << 8) | (img[..., 2] << 16)
    blocks = ticks[::8, ::8]  # one wave per 8x8 pixel block
    mine = blocks[blocks > 0].astype(np.float64)  # ticks
    order = np.sort(mine)
    q = lambda f: float(order[min(len(order) - 1, int(f * len(order)))])
    row = {"scene": name, "parts": parts, "waves": int(mine.size), "kernel_ms": st.kernel_ms,
           "wave_ticks": {"mean": float(mine.mean()), "p50": q(0.5), "p90": q(0.9), "p99": q(0.99), "max": float(order[-1])},
           "sum_of_wave_ticks_over_4096_slots": float(mine.sum() / 4096),
           "waves_over_50k_ticks": int((mine > 50000).sum()), "waves_over_100k_ticks": int((mine > 100000).sum())}
    ys, xs = np.unravel_index(np.argsort(blocks, axis=None)[-5:], blocks.shape)
    row["slowest_blocks_xy"] = [[int(x) * 8, int(y) * 8, float(blocks[y, x])] for y, x in zip(ys, xs)]
    print(json.dumps(row), flush=True)
    if parts == 1:
        np.save(os.path.join(ROOT, "gpurun_out", "wave_%s_%s.npy" % (os.environ.get("RT_AMD_LIB", "x").split("_")[-1].split(".")[0], name.split(".")[0])), blocks.astype(np.int32))
