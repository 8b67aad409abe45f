set -o pipefail
O=gpurun_out/s2o; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "== $*"; env "$@" timeout -k 10 300 python -u tools/bench_configs.py --only $CFG --sweep-frames 30 --counters 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); p=d['per_ray']; print('  %-30s %8.3f ms kernel %8.3f  nodes %.2f tris %.2f quads %.2f' % (d['config'], d['ms_per_frame'], d['kernel_ms_per_frame'], p['node_visits'], p['tri_tests'], p['quadric_tests']))" || exit 1; }
CFG=primitives,glass
run RT_AMD_X=0
CFG=teapot,heightfield
run RT_AMD_TRI_LEAF=4
run RT_AMD_TRI_LEAF=2
run RT_AMD_TRI_LEAF=3
run RT_AMD_TRI_LEAF=4 RT_AMD_SAH_NODE_TRI=1.0
run RT_AMD_TRI_LEAF=4 RT_AMD_SAH_NODE_TRI=0.5
