for t in 4 2 1 8; do echo "TRI_LEAF=$t"; RT_AMD_TRI_LEAF=$t timeout -k 10 200 python -u tools/bench_configs.py --only heightfield,teapot --counters 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  %-30s %8.3f ms kernel %8.3f  %s' % (d['config'], d['ms_per_frame'], d['kernel_ms_per_frame'], d.get('per_ray')))"; done
for t in 2 1 4; do echo "QUAD_LEAF=$t"; RT_AMD_QUAD_LEAF=$t timeout -k 10 200 python -u tools/bench_configs.py --only primitives --sweep-frames 6 --counters 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  %-30s %8.3f ms kernel %8.3f  %s' % (d['config'], d['ms_per_frame'], d['kernel_ms_per_frame'], d.get('per_ray')))"; done
