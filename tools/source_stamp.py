#!/usr/bin/env python3
"""One checksum over the sources the kernels are compiled from (ray-tracer_amd/csrc/*.h, *.hip): counter files under
profiles/ carry it, and bench.py quotes them only while it still matches the tree."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_sources_sha256(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "ray-tracer_amd", "csrc", "*.h")) + glob.glob(os.path.join(root, "ray-tracer_amd", "csrc", "*.hip")))
    for path in files:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_sha256())
