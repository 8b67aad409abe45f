set -o pipefail
O=gpurun_out/s3b; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 600 python -u tools/bench_configs.py --sweep-frames 60 > $O/configs.jsonl 2>>$O/err.log || exit 1
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
