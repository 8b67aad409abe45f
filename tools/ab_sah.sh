run() { echo "== $*"; env "$@" timeout -k 10 300 python -u tools/bench_configs.py --only $CFG --sweep-frames 6 --frames 20 --counters 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); p=d['per_ray']; print('  %-30s kernel %8.3f  nodes %.2f tris %.2f quads %.2f' % (d['config'], d['kernel_ms_per_frame'], p['node_visits'], p['tri_tests'], p['quadric_tests']))" || exit 1; }
CFG=teapot,heightfield
L=$PWD/ray-tracer_amd/lib/librt_amd_s40.so
run RT_AMD_TRI_LEAF=4
run RT_AMD_LIB=$L RT_AMD_TRI_LEAF=4
run RT_AMD_LIB=$L RT_AMD_TRI_LEAF=4 RT_AMD_SAH_NODE_TRI=1.0
run RT_AMD_LIB=$L RT_AMD_TRI_LEAF=4 RT_AMD_SAH_NODE_TRI=1.5
run RT_AMD_LIB=$L RT_AMD_TRI_LEAF=8 RT_AMD_SAH_NODE_TRI=1.5
run RT_AMD_LIB=$L RT_AMD_TRI_LEAF=2
RT_AMD_DEBUG=1 RT_AMD_LIB=$L RT_AMD_TRI_LEAF=4 RT_AMD_SAH_NODE_TRI=1.0 timeout -k 10 100 python -u tools/bench_configs.py --only teapot,heightfield --frames 2 2>&1 | grep "stack_need"
