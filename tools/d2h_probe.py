#!/usr/bin/env python3
"""Probe: device-to-host copy rate of one 1080p rgb8 frame into pinned memory, as one copy on one
stream and as halves on two streams (do two DMA engines beat one over the same PCIe link?)."""
import time
import torch

n = 1920 * 1080 * 3
dev = torch.device("cuda", 0)
src = torch.zeros(n, dtype=torch.uint8, device=dev)
dst = torch.empty(n, dtype=torch.uint8).pin_memory()
s = [torch.cuda.Stream() for _ in range(4)]


def run(parts, reps=200):
    cuts = [n * i // parts for i in range(parts + 1)]
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for p in range(parts):
                with torch.cuda.stream(s[p]):
                    dst[cuts[p]:cuts[p + 1]].copy_(src[cuts[p]:cuts[p + 1]], non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
    return dt


for parts in (1, 2, 4):
    dt = run(parts)
    print("parts %d: %.1f us per frame, %.1f GB/s" % (parts, dt * 1e6, n / dt / 1e9))
