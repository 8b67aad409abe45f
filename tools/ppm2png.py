#!/usr/bin/env python3
"""Converts an ASCII P3 PPM (the renderer's output format) to PNG for viewing."""
import struct
import sys
import zlib

import numpy as np


def read_ppm(path):
    tok = open(path).read().split()
    assert tok[0] == "P3"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.uint8).reshape(h, w, 3)


def write_png(path, img):
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


if __name__ == "__main__":
    write_png(sys.argv[2], read_ppm(sys.argv[1]))
