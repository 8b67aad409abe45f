#!/usr/bin/env python3
"""Runs bench.py once per library variant (lib/librt_amd_<tag>.so) and prints frame and kernel times."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tags = sys.argv[1:]
for rounds in range(2):
    for tag in tags:
        env = dict(os.environ)
        if tag != "default":
            env["RT_AMD_LIB"] = "librt_amd_%s.so" % tag
            env["RT_HOST_LIB"] = "librt_host_%s.so" % tag
        env["RT_AMD_DEBUG"] = "1"
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "3", "--no-cpu-baseline"],
                           env=env, capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            print(tag, "FAILED", p.stderr[-400:])
            sys.exit(1)
        d = json.loads(p.stdout.strip().splitlines()[-1])
        dbg = [l for l in p.stderr.splitlines() if "[rt_amd]" in l]
        print("%-12s frame %.3f ms  kernel %.3f ms  %.2f Grays/s  %s" % (
            tag, d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"] / 1e9, dbg[0] if dbg and rounds == 0 else ""), flush=True)
