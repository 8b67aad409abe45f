#!/usr/bin/env python3
"""Kernel timeline from a rocprofv3 --kernel-trace CSV: for the last launches of the generation kernel
their start, end and duration, the queue they ran on and how many other launches of the kernel were
resident when each started; then the means.  Shows whether consecutive frames overlap on the GPU.

usage: timeline.py <dir with *kernel_trace.csv> [first launch, default: the last 24] [launches, default 24] > table.txt
"""
import csv
import glob
import os
import sys

root = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else None
keep = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = []
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_generation" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Agent_Id", "?")))
rows.sort()
if not rows:
    sys.exit("no k_generation launch in " + root)
if first is None:
    first = max(0, len(rows) - keep)
tail = rows[first:first + keep]
t0 = tail[0][0]
print("# launches %d..%d of %d k_generation launches; times in microseconds from the first listed start" % (
    first, first + len(tail) - 1, len(rows)))
print("# %5s %10s %10s %9s %8s %8s %s" % ("n", "start", "end", "duration", "period", "resident", "queue"))
durs, periods, overlapped = [], [], 0.0
for i, (a, b, q, agent) in enumerate(tail):
    resident = sum(1 for (c, d, _, g) in tail if c < a < d and g == agent)
    period = (a - tail[i - 1][0]) / 1e3 if i else float("nan")
    durs.append((b - a) / 1e3)
    if i:
        periods.append(period)
        overlapped += max(0, min(b, tail[i - 1][1]) - a) / 1e3
    print("  %5d %10.1f %10.1f %9.1f %8.1f %8d %s" % (first + i, (a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3,
                                                   period, resident, q))
span = (tail[-1][1] - tail[0][0]) / 1e3
print("# mean duration %.1f us, mean start-to-start period %.1f us, %d launches in %.1f us = %.1f us per launch" % (
    sum(durs) / len(durs), sum(periods) / max(1, len(periods)), len(tail), span, span / len(tail)))
print("# time two consecutive launches were resident together: %.1f us of %.1f us (%.0f %%)" % (
    overlapped, span, 100.0 * overlapped / span))
