#!/bin/bash
# Frame times of the triangle configurations against the host builder's knobs (longest triangle run per
# leaf, cost of a node visit relative to a triangle test).  usage: tools/leaf_sweep.sh <out-dir>
set -o pipefail
O=$1
mkdir -p $O
: > $O/leaf_sweep.txt
for leaf in 4 2 3 6 8; do
  for cost in -1 0.5 1.0 2.0; do
    RT_AMD_TRI_LEAF=$leaf RT_AMD_SAH_NODE_TRI=$cost timeout -k 10 200 python3 -u tools/bench_configs.py --only teapot,heightfield --frames 20 --counters > $O/sweep.jsonl 2> $O/sweep.err || { echo "failed at $leaf $cost"; tail -5 $O/sweep.err; exit 1; }
    python3 - $leaf $cost $O/sweep.jsonl <<'E' | tee -a $O/leaf_sweep.txt
import json, sys
line = "run<=%s node/tri cost %s:" % (sys.argv[1], sys.argv[2])
for l in open(sys.argv[3]):
    d = json.loads(l)
    line += "  %s %.3f ms (nodes %.2f tris %.2f per ray)" % (d["config"].split()[0], d["kernel_ms_per_frame"], d["per_ray"]["node_visits"], d["per_ray"]["tri_tests"])
print(line, flush=True)
E
  done
done
