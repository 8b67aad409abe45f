#!/usr/bin/env python3
"""ms/frame and rays/s of every measured configuration on one GPU (not the headline bench: bench.py).

  example 1080p, glass 4K, pattern 960x540, teapot 1080p, 1M-triangle heightfield 4K,
  100k primitives 8K (camera sweep, --sweep-frames frames)

Prints one JSON line per configuration.  The frame stays in HBM (rt_render into a device buffer).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scenegen  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


FP64_VALU_PEAK, HBM_PEAK = 78.6e12, 8.0e12
SURVEY = {"node": 22, "tri": 46, "quad": 60, "hit": 150}  # flops per event, the measurement plan's coefficients
WARM = 8


def roofline(c, kernel_ms, scene_bytes):
    """Events of one instrumented frame priced with the plan's coefficients, and the HBM bytes the
    algorithm cannot avoid (scene once per generation, 3 B per pixel, 120 B written and read back per
    recorded ray), against the generation kernels' time of one frame."""
    nv, tt = c.node_visits + c.shadow_node_visits, c.tri_tests + c.shadow_tri_tests
    qt, gt = c.quadric_tests + c.shadow_quadric_tests, c.gate_tests + c.shadow_gate_tests
    flops = nv * SURVEY["node"] + tt * SURVEY["tri"] + qt * SURVEY["quad"] + c.shadow * SURVEY["hit"]
    secondary = c.reflect + c.refract
    records = (c.primary if secondary else 0) + secondary
    hbm = c.kernel_launches * scene_bytes + c.primary * 3 + 2 * records * 120
    t = kernel_ms * 1e-3
    return {"events": {"node_visits": nv, "tri_tests": tt, "quadric_tests": qt, "gate_tests": gt, "shaded_hits": c.shadow},
            "flops_per_frame": flops, "unavoidable_hbm_bytes_per_frame": hbm, "kernel_ms_per_frame": kernel_ms,
            "frac_fp64_valu": flops / t / FP64_VALU_PEAK, "frac_hbm_unavoidable": hbm / t / HBM_PEAK,
            "launches_per_frame": c.kernel_launches}


def run(pkg, name, path, size, frames, cameras=None, counters=False):
    t0 = time.perf_counter()
    scene = pkg.Scene.from_file(path, base_dir=ROOT, device=0)
    load_s = time.perf_counter() - t0
    if size:
        scene.set_canvas(*size)
    info, foot = scene.info(), scene.footprint()
    w, h = info["width"], info["height"]
    frame = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
    for k in range(WARM):  # first touches of freshly allocated ray queues, clocks
        scene.render_device(rgb8_ptr=frame.data_ptr())
    torch.cuda.synchronize()
    kernel_ms = render_ms = 0.0
    rays = 0
    t0 = time.perf_counter()
    for k in range(frames):
        if cameras:
            scene.set_camera(w, h, 1.0472, *cameras[k % len(cameras)], [0.0, 1.0, 0.0])
        st = scene.render_device(rgb8_ptr=frame.data_ptr())
        kernel_ms += st.kernel_ms
        render_ms += st.render_ms
        rays += st.rays
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # the same frames only enqueued (RT_FLAG_ASYNC) into two device frames in turn, collected at the end: what a
    # caller gets who does not wait for every frame (frames of scenes that spawn rays are verified as they go)
    frames2 = [frame, torch.zeros_like(frame)]
    for k in range(2):
        scene.render_device(rgb8_ptr=frames2[k].data_ptr(), wait=False)
    scene.wait()
    t0 = time.perf_counter()
    for k in range(frames):
        if cameras:
            scene.set_camera(w, h, 1.0472, *cameras[k % len(cameras)], [0.0, 1.0, 0.0])
        scene.render_device(rgb8_ptr=frames2[k % 2].data_ptr(), wait=False)
    enq = scene.wait()
    wall_enqueued = time.perf_counter() - t0
    assert enq.rays == rays, (enq.rays, rays)
    out = {"config": name, "size": [w, h], "frames": frames, "ms_per_frame": wall / frames * 1e3,
           "ms_per_frame_enqueued": wall_enqueued / frames * 1e3,
           "kernel_ms_per_frame": kernel_ms / frames, "rays_per_frame": rays / frames, "rays_per_s": rays / wall,
           "generations": st.generations, "leaves": info["leaves"], "triangles": info["triangles"],
           "scene_MB": foot["scene_bytes"] / 1e6, "bvh_build_s": foot["build_seconds"], "load_s": load_s}
    if counters:
        c = scene.render_device(rgb8_ptr=frame.data_ptr(), counters=True)
        out["roofline"] = roofline(c, kernel_ms / frames, foot["scene_bytes"])
        out["per_ray"] = {"node_visits": (c.node_visits + c.shadow_node_visits) / c.rays,
                          "tri_tests": (c.tri_tests + c.shadow_tri_tests) / c.rays,
                          "quadric_tests": (c.quadric_tests + c.shadow_quadric_tests) / c.rays,
                          "gate_tests": (c.gate_tests + c.shadow_gate_tests) / c.rays}
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--sweep-frames", type=int, default=60)
    ap.add_argument("--only", default="")
    ap.add_argument("--counters", action="store_true")
    ap.add_argument("--warm", type=int, default=8, help="untimed frames before the timed ones")
    args = ap.parse_args()
    global WARM
    WARM = args.warm
    pkg = load_package()
    want = set(args.only.split(",")) if args.only else None

    def on(name):
        return want is None or name in want

    sc = os.path.join(ROOT, "scenes")
    if on("example"):
        run(pkg, "example 1080p", os.path.join(sc, "example.json"), None, args.frames, counters=args.counters)
    if on("glass"):
        run(pkg, "glass 4K", os.path.join(sc, "glass.json"), None, args.frames, counters=args.counters)
    if on("pattern"):
        run(pkg, "pattern 960x540", os.path.join(sc, "pattern.json"), None, args.frames, counters=args.counters)
    if on("teapot"):
        run(pkg, "teapot 1080p", os.path.join(sc, "teapot.json"), (1920, 1080), args.frames, counters=args.counters)
    gen = tempfile.mkdtemp(prefix="rtgen")
    if on("heightfield"):
        path, tris = scenegen.heightfield(gen, cells=708)
        run(pkg, "heightfield 1M triangles 4K", path, None, args.frames, counters=args.counters)
    if on("primitives"):
        path, n = scenegen.primitives(gen, count=100000)
        cams = [scenegen.sweep_camera(k) for k in range(60)]
        run(pkg, "100k primitives 8K sweep", path, None, args.sweep_frames, cameras=cams, counters=args.counters)


if __name__ == "__main__":
    main()
