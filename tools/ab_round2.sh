set -o pipefail
O=gpurun_out/s2l; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for v in product trikept product trikept; do
  if [ $v = trikept ]; then export RT_AMD_LIB=librt_amd_trikept.so RT_HOST_LIB=librt_host_trikept.so; else unset RT_AMD_LIB RT_HOST_LIB; fi
  timeout -k 10 200 python -u bench.py --no-cpu-baseline > $O/bench_$v.json 2>>$O/err.log || exit 1
  timeout -k 10 600 python -u tools/bench_configs.py --only teapot,heightfield > $O/configs_$v.jsonl 2>>$O/err.log || exit 1
  python - <<P
import json
d=json.loads(open("$O/bench_$v.json").read().strip().splitlines()[-1]); print("$v bench ms/step %.4f"%d["ms_per_step"],"hbm %.4f"%d["frames_left_in_hbm"]["ms_per_step"],"lone %.4f"%d["roofline"]["lone_launch_ms"], d["frame_sha256"][:8])
for l in open("$O/configs_$v.jsonl"):
    d=json.loads(l); print("$v", d["config"], "%.3f ms"%d["ms_per_frame"], "kernel %.3f"%d["kernel_ms_per_frame"])
P
done
