# A/B of the frame loop's copy ordering on one GPU: fence stride x deferred hand-over (tools, not part of the bench)
set -o pipefail
O=gpurun_out/s2c; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_loop.py tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for cfg in "1 0" "2 0" "1 1" "2 1" "3 1" "2 0" "2 1"; do set -- $cfg
  RT_AMD_COPIES_PER_FENCE=$1 RT_AMD_DEFERRED_COPY=$2 timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/b20_s$1_d$2.json 2>>$O/err.log || exit 1
  RT_AMD_COPIES_PER_FENCE=$1 RT_AMD_DEFERRED_COPY=$2 timeout -k 10 200 python -u bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/b200_s$1_d$2.json 2>>$O/err.log || exit 1
  python - <<P
import json
for k in (20,200):
    d=json.loads(open("$O/b%d_s$1_d$2.json"%k).read().strip().splitlines()[-1])
    print("stride $1 deferred $2 steps",k,"ms/step %.4f"%d["ms_per_step"],"hbm %.4f"%d["frames_left_in_hbm"]["ms_per_step"],"lone %.4f"%d["roofline"]["lone_launch_ms"], d["frame_sha256"][:8])
P
done
