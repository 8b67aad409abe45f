#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one run each) into per-launch HBM bytes
of the generation kernel:  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies a 128-B request as 64 B for
16-B-per-lane loads (all record loads of the traversal are 16-B loads), hence the factor 2; the
factor is calibrated for streaming reads, so for these scattered reads it is an upper bound.

usage: traffic_json.py <dir with p1/ p2/ rocprofv3 outputs> <out.json>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_stamp import kernel_sources_sha256  # noqa: E402

root, out = sys.argv[1], sys.argv[2]
tot = defaultdict(float)
disp = defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_generation" not in row["Kernel_Name"]:
            continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        disp[row["Counter_Name"]].add(row["Dispatch_Id"])
fetch = tot["FETCH_SIZE"] / max(1, len(disp["FETCH_SIZE"]))
write = tot["WRITE_SIZE"] / max(1, len(disp["WRITE_SIZE"]))
res = {"k_generation": {
    "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
    "launches_seen": [len(disp["FETCH_SIZE"]), len(disp["WRITE_SIZE"])],
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024; separate --pmc passes; gfx950 FETCH_SIZE halving corrected",
}, "kernel_sources_sha256": kernel_sources_sha256()}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
