#!/bin/bash
# A/B of kernel library variants (lib/librt_amd_<tag>.so, `default` = the product build) on one box:
# the pipelined teapot loop with the frames left in HBM, and the blocking per-configuration frames.
# usage: tools/ab_configs.sh <out-dir> <tag>...      every GPU step under its own time limit
set -o pipefail
O=$1; shift
mkdir -p $O
for tag in "$@"; do
  if [ "$tag" = default ]; then unset RT_AMD_LIB RT_HOST_LIB; else export RT_AMD_LIB=librt_amd_$tag.so RT_HOST_LIB=librt_host_$tag.so; fi
  timeout -k 10 200 python3 -u bench.py --steps 200 --warmup 5 --no-cpu-baseline --hbm-only > $O/bench_$tag.json 2> $O/bench_$tag.err || { echo "bench $tag failed"; tail -5 $O/bench_$tag.err; exit 1; }
  timeout -k 10 400 python3 -u tools/bench_configs.py --only ${AB_ONLY:-teapot,glass,heightfield,primitives} --frames 20 --sweep-frames 12 > $O/configs_$tag.jsonl 2> $O/configs_$tag.err || { echo "configs $tag failed"; tail -5 $O/configs_$tag.err; exit 1; }
  python3 - $tag $O <<'E'
import json, sys
tag, o = sys.argv[1], sys.argv[2]
b = json.loads(open("%s/bench_%s.json" % (o, tag)).read().strip().splitlines()[-1])
h = b.get("frames_left_in_hbm") or {}
line = "%-8s hbm loop %.4f ms lone %.4f ms |" % (tag, h.get("ms_per_step", b["ms_per_step"]), b["roofline"]["lone_launch_ms"])
for l in open("%s/configs_%s.jsonl" % (o, tag)):
    d = json.loads(l)
    line += " %s %.3f" % (d["config"].split()[0], d["kernel_ms_per_frame"])
print(line, flush=True)
E
done
