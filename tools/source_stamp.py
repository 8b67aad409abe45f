#!/usr/bin/env python3
"""One checksum over the sources the kernels are compiled from: every ray-tracer_amd/csrc/*.hip and every file they
include, directly or through other includes (quoted includes, followed from the including file's directory; the
host-only headers and .cpp files of the library are not among them), with comments and blank space taken out, so
that a reworded comment in the public header does not make measured counters look stale.  Counter files under
profiles/ carry it, and bench.py quotes them only while it still matches the tree."""
import glob
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = re.compile(rb'^\s*#\s*include\s*"([^"]+)"', re.M)
# a string or character literal (kept), or a comment (dropped)
TOKEN = re.compile(rb'"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'|//[^\n]*|/\*.*?\*/', re.S)


def code_only(text):
    text = TOKEN.sub(lambda m: m.group(0) if m.group(0)[:1] in (b'"', b"'") else b" ", text)
    return b"\n".join(b" ".join(line.split()) for line in text.splitlines() if line.strip())


def kernel_sources(root=ROOT):
    todo = sorted(glob.glob(os.path.join(root, "ray-tracer_amd", "csrc", "*.hip")))
    seen = {}
    while todo:
        path = os.path.normpath(todo.pop())
        if path in seen or not os.path.isfile(path):
            continue
        seen[path] = open(path, "rb").read()
        for name in INCLUDE.findall(seen[path]):
            todo.append(os.path.join(os.path.dirname(path), name.decode()))
    return seen


def kernel_sources_sha256(root=ROOT):
    h = hashlib.sha256()
    files = kernel_sources(root)
    for path in sorted(files):
        h.update(os.path.relpath(path, root).encode())
        h.update(code_only(files[path]))
    return h.hexdigest()


if __name__ == "__main__":
    for path in sorted(kernel_sources()):
        print("#", os.path.relpath(path, ROOT))
    print(kernel_sources_sha256())
