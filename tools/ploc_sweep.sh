run() { echo "== $*"; env "$@" RT_AMD_DEVICE_BVH=1 timeout -k 10 300 python -u tools/bench_configs.py --only heightfield,teapot --frames 10 --counters 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); p=d['per_ray']; print('  %-30s kernel %8.3f  nodes %.2f tris %.2f' % (d['config'], d['kernel_ms_per_frame'], p['node_visits'], p['tri_tests']))" || exit 1; }
run RT_AMD_PLOC_RADIUS=4
run RT_AMD_PLOC_RADIUS=12
run RT_AMD_PLOC_RADIUS=32
run RT_AMD_PLOC_RADIUS=100
run RT_AMD_PLOC_RADIUS=12 RT_AMD_DEVICE_TRI_RUN=6
run RT_AMD_PLOC_RADIUS=12 RT_AMD_DEVICE_TRI_RUN=8
run RT_AMD_PLOC_RADIUS=12 RT_AMD_DEVICE_TRI_RUN=2
RT_AMD_DEBUG=1 RT_AMD_DEVICE_BVH=0 timeout -k 10 100 python -u tools/bench_configs.py --only heightfield --frames 2 2>&1 | grep "nodes="
RT_AMD_DEBUG=1 RT_AMD_DEVICE_BVH=1 timeout -k 10 100 python -u tools/bench_configs.py --only heightfield --frames 2 2>&1 | grep "nodes="
