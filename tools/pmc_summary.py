#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel and counter (mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"].split("(")[0][:60]
        key = (name, row["Counter_Name"])
        acc[key][0] += float(row["Counter_Value"])
        acc[key][1] += 1
# a dispatch appears once per (dimension instance) row; report totals divided by dispatch count estimate
disp = defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        disp[row["Kernel_Name"].split("(")[0][:60]].add(row["Dispatch_Id"])
for (name, ctr), (total, rows) in sorted(acc.items()):
    if "k_generation" not in name:
        continue
    n = max(1, len(disp[name]))
    print("%-50s %-28s per-dispatch %.4g  (dispatches %d)" % (name, ctr, total / n, n))
