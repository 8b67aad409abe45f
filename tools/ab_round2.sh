set -o pipefail
O=gpurun_out/s2j; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for n in 4096 2048 1024; do
  RT_AMD_BAND_GRID_MAX=$n timeout -k 10 300 python -u tools/bench_configs.py --only primitives --sweep-frames 30 > $O/prim_$n.jsonl 2>>$O/err.log || exit 1
  python - <<P
import json
d=json.loads(open("$O/prim_$n.jsonl").read().strip().splitlines()[-1]); print("grid $n", "%.3f ms"%d["ms_per_frame"], "kernel %.3f"%d["kernel_ms_per_frame"], "scene %.0f MB"%d["scene_MB"], "load %.2f s"%d["load_s"])
P
done
timeout -k 10 300 python -u tools/bench_configs.py --only example,glass,pattern,teapot,heightfield > $O/configs.jsonl 2>>$O/err.log || exit 1
python - <<P
import json
for l in open("$O/configs.jsonl"):
    d=json.loads(l); print(d["config"], "%.3f ms"%d["ms_per_frame"], "kernel %.3f"%d["kernel_ms_per_frame"])
P
timeout -k 10 200 python -u bench.py --no-cpu-baseline > $O/bench200.json 2>>$O/err.log || exit 1
python - <<P
import json
d=json.loads(open("$O/bench200.json").read().strip().splitlines()[-1]); print("bench ms/step %.4f"%d["ms_per_step"],"hbm %.4f"%d["frames_left_in_hbm"]["ms_per_step"],"lone %.4f"%d["roofline"]["lone_launch_ms"])
P
