#!/usr/bin/env python3
"""Canvas::save three ways on a rendered frame: the host writer (one core), rt_ppm_encode on a frame
that is already in HBM (text left in HBM), and rt_ppm_encode_host (frame up, text down)."""
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    pkg = load_package()
    b = pkg.bindings
    for name, size in (("example", None), ("glass", None), ("glass", (7680, 4320))):
        scene = pkg.Scene.from_file(os.path.join(ROOT, "scenes", name + ".json"), base_dir=ROOT, device=0)
        if size:
            scene.set_canvas(*size)
        info = scene.info()
        w, h = info["width"], info["height"]
        frame = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda:0")
        scene.render_device(rgb8_ptr=frame.data_ptr())
        cap = b.ppm_max_bytes(w, h)
        out = torch.empty((cap,), dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        n = b.ppm_encode_device(frame.data_ptr(), w, h, out.data_ptr(), cap)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            n = b.ppm_encode_device(frame.data_ptr(), w, h, out.data_ptr(), cap)
        dev_ms = (time.perf_counter() - t0) / reps * 1e3
        host_frame = frame.cpu().numpy()
        text = b.ppm_encode(host_frame)
        _, host = b.libraries()
        buf = ctypes.create_string_buffer(cap)
        t0 = time.perf_counter()  # one pass of the host writer (the binding above runs it twice: size, then text)
        host.rth_ppm_encode(w, h, host_frame.ctypes.data_as(ctypes.c_void_p), buf, cap)
        host_ms = (time.perf_counter() - t0) * 1e3
        b.ppm_encode_via_device(host_frame)
        t0 = time.perf_counter()
        text2 = b.ppm_encode_via_device(host_frame)
        staged_ms = (time.perf_counter() - t0) * 1e3
        assert text == text2 and len(text) == n
        print(json.dumps({"scene": name, "size": [w, h], "file_MB": n / 1e6, "host_writer_ms": host_ms,
                          "device_ms": dev_ms, "device_GBps_of_text": n / dev_ms / 1e6,
                          "through_host_buffers_ms": staged_ms}), flush=True)


if __name__ == "__main__":
    main()
