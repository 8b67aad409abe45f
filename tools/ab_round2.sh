set -o pipefail
O=gpurun_out/s2p; mkdir -p $O
D="RT_AMD_LIB=librt_amd_stamps.so RT_HOST_LIB=librt_host_stamps.so"
for lv in 4 3; do
  echo "== passes $lv"
  env $D RT_AMD_TIMING_BAND_PASSES=$lv timeout -k 10 300 python -u tools/bench_configs.py --only primitives --sweep-frames 4 --warm 2 > $O/prim_$lv.jsonl 2> $O/prim_$lv.err || exit 1
  grep "rt_amd" $O/prim_$lv.err | tail -12
done
