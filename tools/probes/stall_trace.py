#!/usr/bin/env python3
"""Reads a rocprofv3 --hip-trace --kernel-trace run (csv) of tools/probes/stall_probe.py and names what a start-up
stall of the frame loop is spent in: every HIP call longer than a threshold, with the time since the first
render kernel and the number of render kernels launched before it, and every gap between consecutive render
kernels longer than the threshold.  usage: stall_trace.py <rocprofv3 output dir> [threshold ms]"""
import csv
import glob
import os
import sys

root = sys.argv[1]
thr = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.5e6  # ns


def rows(pattern):
    for path in glob.glob(os.path.join(root, "**", pattern), recursive=True):
        for r in csv.DictReader(open(path)):
            yield r


kernels = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows("*kernel_trace.csv")
                  if "k_generation" in r["Kernel_Name"]), key=lambda k: k[0])
if not kernels:
    sys.exit("no k_generation launch in " + root)
t0 = kernels[0][0]
starts = [k[0] for k in kernels]
print("# %d render kernels over %.1f ms; threshold %.2f ms" % (len(kernels), (kernels[-1][1] - t0) / 1e6, thr / 1e6))
print("# gaps between the end of a render kernel and the start of the next one")
for a, b in zip(kernels, kernels[1:]):
    if b[0] - a[1] > thr:
        print("  after kernel %5d at %9.3f ms: idle %.3f ms" % (starts.index(a[0]), (a[1] - t0) / 1e6, (b[0] - a[1]) / 1e6))
print("# HIP calls longer than the threshold (after the first render kernel)")
import bisect
for r in sorted(rows("*hip_api_trace.csv"), key=lambda r: int(r["Start_Timestamp"])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e - s > thr and s > t0:
        print("  %-34s at %9.3f ms for %8.3f ms (render kernels launched before it: %d, thread %s)" % (
            r["Function"], (s - t0) / 1e6, (e - s) / 1e6, bisect.bisect_left(starts, s), r["Thread_Id"]))
