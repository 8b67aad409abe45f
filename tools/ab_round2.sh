set -o pipefail
O=gpurun_out/s2m; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_band_grid.py tests/test_gpu_fuzz.py tests/test_gpu_scale.py tests/test_gpu_golden.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for n in 1024 2048 4096 512; do
  RT_AMD_BAND_GRID_MAX=$n timeout -k 10 300 python -u tools/bench_configs.py --only primitives --sweep-frames 30 > $O/prim_$n.jsonl 2>>$O/err.log || exit 1
  python - <<P
import json
d=json.loads(open("$O/prim_$n.jsonl").read().strip().splitlines()[-1]); print("grid $n", "%.3f ms"%d["ms_per_frame"], "kernel %.3f"%d["kernel_ms_per_frame"], "scene %.0f MB"%d["scene_MB"], "load %.2f s"%d["load_s"])
P
done
