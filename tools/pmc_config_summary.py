#!/usr/bin/env python3
"""Per-frame counter totals of the ray-generation kernels of one configuration and what they say about
which ceiling binds.  Input: a directory of rocprofv3 --pmc passes (tools/pmc_configs.sh), the number of
frames each pass rendered, the configuration's name.

HBM bytes per frame = (2 FETCH_SIZE + WRITE_SIZE) KiB (FETCH_SIZE tallies a 128-B request as 64 B for the
16-B-per-lane loads this kernel issues: an upper bound for scattered reads).  SQ cycle counters tick once
per 4 clocks; SQ_WAVE_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs / 4) = waves in flight per SIMD.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root, frames, name = sys.argv[1], int(sys.argv[2]), sys.argv[3]
tot = defaultdict(float)
launches = defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_generation" not in row["Kernel_Name"]:
            continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        launches[row["Counter_Name"]].add((path, row["Dispatch_Id"]))
f = {k: v / frames for k, v in tot.items()}
g = lambda k: f.get(k, 0.0)
print("# %s: rocprofv3 --pmc, seven separate passes over `python3 tools/bench_configs.py --only %s`, %d frames per pass;" % (name, name, frames))
print("# totals over all k_generation launches of one frame (%d launches per frame), summed over the 8 XCDs" % (
    len(launches.get("SQ_WAVES", launches.get("FETCH_SIZE", set()))) // frames))
for k in sorted(f):
    print("%-32s %.4g" % (k, f[k]))
hbm = (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0
gui = g("GRBM_GUI_ACTIVE") / 8.0  # per XCD: busy clocks of the frame's generation kernels (profiled, serialised)
secs = gui / 2.4e9 if gui else 0.0
f64i = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
f64ops = 64.0 * (g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + 2.0 * g("SQ_INSTS_VALU_FMA_F64"))
print("# derived")
print("# HBM traffic per frame                 = %.1f MB  (fetch %.1f MiB x 2, write %.1f MiB)" % (hbm / 1e6, g("FETCH_SIZE") / 1024, g("WRITE_SIZE") / 1024))
if secs:
    print("# kernel busy time per frame (profiled)  = %.3f ms at 2.4 GHz" % (secs * 1e3))
    print("# HBM bandwidth over that time           = %.1f GB/s = %.3f of 8 TB/s" % (hbm / secs / 1e9, hbm / secs / 8e12))
    print("# L2 -> fabric read requests             = %.3g x 64 B = %.1f MB per frame, %.1f GB/s" % (
        g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_sum") * 64 / 1e6, g("TCC_EA0_RDREQ_sum") * 64 / secs / 1e9))
    print("# L2 requests                            = %.3g per frame x 128 B = %.1f GB/s of L2 bandwidth" % (
        g("TCC_REQ_sum"), g("TCC_REQ_sum") * 128 / secs / 1e9))
    print("# f64 operations issued                  = %.3g per frame = %.2f TFLOP/s = %.3f of 78.6 (64 lanes; fma twice)" % (
        f64ops, f64ops / secs / 1e12, f64ops / secs / 78.6e12))
    if g("SQ_INSTS_VALU"):
        print("# VALU issue busy                        = %.3f  (4 clocks per wave instruction over 1024 SIMDs)" % (
            4.0 * g("SQ_INSTS_VALU") / (gui * 1024.0)))
    if g("SQ_WAVE_CYCLES"):
        print("# waves in flight per SIMD               = %.2f" % (4.0 * g("SQ_WAVE_CYCLES") / (gui * 1024.0)))
    if g("TD_TD_BUSY_sum"):
        print("# TD busy / TA busy (mean over 256 CUs)  = %.3f / %.3f of the kernel time" % (
            g("TD_TD_BUSY_sum") / 256.0 / gui, g("TA_TA_BUSY_sum") / 256.0 / gui))
if g("TCC_REQ_sum"):
    print("# L2 hit rate                            = %.3f" % (g("TCC_HIT_sum") / g("TCC_REQ_sum")))
if g("TCP_TOTAL_CACHE_ACCESSES_sum"):
    print("# vector L1 hit rate                     = %.4f" % (1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")))
if g("SQ_INSTS_VALU"):
    print("# f64 share of VALU instructions         = %.3f" % (f64i / g("SQ_INSTS_VALU")))
if g("SQ_ACTIVE_INST_VALU"):
    print("# active lanes per VALU instruction      = %.3f" % (g("SQ_THREAD_CYCLES_VALU") / (64.0 * g("SQ_ACTIVE_INST_VALU"))))
if g("SQ_WAVE_CYCLES"):
    print("# wave cycles: waiting %.3f, issuing %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY") / g("SQ_WAVE_CYCLES")))
