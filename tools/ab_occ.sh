for e in 0 2048 5000 12000 32000; do echo "extra LDS $e"; RT_AMD_EXTRA_LDS=$e timeout -k 10 300 python -u tools/bench_configs.py --only example,teapot,heightfield --frames 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  %-30s %8.3f ms kernel %8.3f' % (d['config'], d['ms_per_frame'], d['kernel_ms_per_frame']))" || exit 1; done
