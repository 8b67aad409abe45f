#!/bin/bash
# One GPU session that collects everything profiles/ holds for a tree: tools/gpu_round.sh (smoke, bench, kernel
# traces with timelines, counter passes, per-part periods, probes, two-rank rehearsal, per-wave figures), the
# default bench line, counters and frame times per configuration, and for the 100k-primitive field the cone
# index's resolution table, its pass-by-pass timing (diagnostic build) and the phase shares with the band hits
# used and dropped.  Needs `make -C ray-tracer_amd all diag probes lib/librt_host_nobands.so lib/librt_host_stamps.so`.
# usage: tools/profile_round.sh [tag]      results under gpurun_out/<tag>/
# RT_PROFILE_SKIP_ROUND=1: tools/gpu_round.sh <tag> ran in a call of its own (together they can exceed one call's limit)
set -o pipefail
tag=${1:-profile}
if [ -z "$RT_PROFILE_SKIP_ROUND" ]; then bash tools/gpu_round.sh $tag || exit 1; fi
O=gpurun_out/$tag
mkdir -p $O
timeout -k 10 300 python -u bench.py > $O/bench_final_tree.json 2> $O/bench_final_tree.err || exit 1
bash tools/pmc_configs.sh $tag teapot glass heightfield primitives > $O/pmc_configs.log 2>&1 || { tail -20 $O/pmc_configs.log; exit 1; }
timeout -k 10 600 python -u tools/bench_configs.py --counters > $O/configs.jsonl 2>>$O/err.log || exit 1
: > $O/primitives_band_grid_resolution.jsonl
for n in 512 1024 2048 4096; do
  RT_AMD_BAND_GRID_MAX=$n timeout -k 10 300 python -u tools/bench_configs.py --only primitives --sweep-frames 60 >> $O/primitives_band_grid_resolution.jsonl 2>>$O/err.log || exit 1
done
: > $O/primitives_band_passes_timing_only.jsonl
D="RT_AMD_LIB=librt_amd_nobands.so RT_HOST_LIB=librt_host_nobands.so"
env $D RT_AMD_TIMING_NO_BANDS=1 timeout -k 10 300 python -u tools/bench_configs.py --only primitives --sweep-frames 60 | sed 's/"config": "/"config": "list switched off: /' >> $O/primitives_band_passes_timing_only.jsonl || exit 1
for lv in 0 1 2 4 3; do
  env $D RT_AMD_TIMING_BAND_PASSES=$lv timeout -k 10 300 python -u tools/bench_configs.py --only primitives --sweep-frames 60 | sed "s/\"config\": \"/\"config\": \"passes $lv: /" >> $O/primitives_band_passes_timing_only.jsonl || exit 1
done
S="RT_AMD_LIB=librt_amd_stamps.so RT_HOST_LIB=librt_host_stamps.so"
env $S timeout -k 10 300 python -u tools/bench_configs.py --only primitives --sweep-frames 2 --warm 1 > /dev/null 2> $O/stamps3.err || exit 1
env $S RT_AMD_TIMING_BAND_PASSES=4 timeout -k 10 300 python -u tools/bench_configs.py --only primitives --sweep-frames 2 --warm 1 > /dev/null 2> $O/stamps4.err || exit 1
grep "rt_amd" $O/stamps3.err | tail -18 > $O/primitives_phase_shares.txt
grep "rt_amd" $O/stamps4.err | tail -18 > $O/primitives_phase_shares_band_hits_dropped.txt
python - <<P
import json
for f in ("configs","primitives_band_grid_resolution","primitives_band_passes_timing_only"):
    for l in open("$O/%s.jsonl"%f):
        d=json.loads(l); print(f[:12], d["config"], "%.3f ms"%d["ms_per_frame"], "kernel %.3f"%d["kernel_ms_per_frame"], "rays %.4g"%d["rays_per_frame"], "%.1f Grays/s"%(d["rays_per_s"]/1e9), "scene %.0f MB"%d["scene_MB"])
P
