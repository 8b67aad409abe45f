set -o pipefail
O=gpurun_out/s2k; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for v in corner tight; do
  if [ $v = corner ]; then export RT_AMD_CORNER_BOXES=1; else unset RT_AMD_CORNER_BOXES; fi
  timeout -k 10 600 python -u tools/bench_configs.py --only example,glass,pattern,primitives --sweep-frames 30 --counters > $O/configs_$v.jsonl 2>>$O/err.log || exit 1
  python - <<P
import json
for l in open("$O/configs_$v.jsonl"):
    d=json.loads(l); print("$v", d["config"], "%.3f ms"%d["ms_per_frame"], "kernel %.3f"%d["kernel_ms_per_frame"], {k:round(x,2) for k,x in d["per_ray"].items()})
P
done
