#!/usr/bin/env python3
"""Registers and scratch of every kernel in gfx950 assembly files (hipcc --cuda-device-only -S): one line per kernel.
usage: tools/kernel_resources.py file.s..."""
import re
import subprocess
import sys

for path in sys.argv[1:]:
    txt = open(path).read()
    for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n\s+\.sgpr_count:\s+(\d+)(?:.|\n)*?\.vgpr_count:\s+(\d+)", txt):
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        print("%-64s scratch %4s B per lane  vgpr %3s  sgpr %3s" % (name, m.group(2), m.group(4), m.group(3)))
