# Frame time and per-ray counts through the hierarchy built on the GPU and the one built on the host, then the GPU build's timing.
run() { echo "== $*"; env "$@" timeout -k 10 300 python -u tools/bench_configs.py --only heightfield,teapot,primitives --frames 10 --counters 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); p=d['per_ray']; print('  %-30s kernel %8.3f  nodes %.2f tris %.2f quads %.2f' % (d['config'], d['kernel_ms_per_frame'], p['node_visits'], p['tri_tests'], p['quadric_tests']))" || exit 1; }
run RT_AMD_DEVICE_BVH=1
run RT_AMD_DEVICE_BVH=0
RT_AMD_DEVICE_BVH=1 RT_AMD_BUILD_TIMING=1 timeout -k 10 200 python -u tools/bench_configs.py --only heightfield,teapot,primitives --frames 2 2>&1 | grep "rt_amd"
