#!/usr/bin/env python3
"""f64 operations the generation kernel issues per launch, from rocprofv3 --pmc passes:
64 lanes x (ADD_F64 + MUL_F64 + 2 FMA_F64) wave instructions (an upper bound: inactive lanes count).
With SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU in the same passes the active-lane share is applied too.

usage: pmc_f64_json.py <dir with rocprofv3 outputs> <out.json>
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
tot, disp = defaultdict(float), defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if "k_generation" not in row["Kernel_Name"]:
            continue
        tot[row["Counter_Name"]] += float(row["Counter_Value"])
        disp[row["Counter_Name"]].add((path, row["Dispatch_Id"]))
per = {k: tot[k] / max(1, len(disp[k])) for k in tot}
add, mul, fma = per.get("SQ_INSTS_VALU_ADD_F64", 0), per.get("SQ_INSTS_VALU_MUL_F64", 0), per.get("SQ_INSTS_VALU_FMA_F64", 0)
valu = per.get("SQ_INSTS_VALU", 0)
lanes = None
if per.get("SQ_ACTIVE_INST_VALU"):
    lanes = per.get("SQ_THREAD_CYCLES_VALU", 0) / (64.0 * per["SQ_ACTIVE_INST_VALU"])
res = {"k_generation": {
    "per_launch": per,
    "f64_wave_instructions_per_launch": add + mul + fma,
    "f64_share_of_valu": (add + mul + fma) / valu if valu else None,
    "f64_ops_per_launch": 64.0 * (add + mul + 2.0 * fma),
    "active_lane_share": lanes,
    "f64_ops_per_launch_active_lanes": None if lanes is None else 64.0 * (add + mul + 2.0 * fma) * lanes,
    "formula": "64 lanes x (ADD_F64 + MUL_F64 + 2 FMA_F64) wave-level instructions, mean per launch",
    # what binds the kernel when it is not arithmetic: the share of issue slots taken (a wave instruction occupies a
    # SIMD for 4 clocks; 1024 SIMDs) and the share of time the vector-memory return path is busy (256 TDs), over
    # the clocks the launches were resident (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
    "valu_issue_busy": None if not per.get("GRBM_GUI_ACTIVE") else valu * 4.0 / (1024.0 * per["GRBM_GUI_ACTIVE"] / 8.0),
    "td_busy": None if not (per.get("GRBM_GUI_ACTIVE") and per.get("TD_TD_BUSY_sum")) else per["TD_TD_BUSY_sum"] / (256.0 * per["GRBM_GUI_ACTIVE"] / 8.0),
}, "kernel_sources_sha256": kernel_sources_sha256()}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["k_generation"], indent=1))
