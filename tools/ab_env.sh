# usage: tools/ab_env.sh "<configs>" "VAR=1" ...   runs bench_configs once per environment setting ("" = none)
cfg=$1; shift
for e in "$@"; do echo "== ${e:-default}"; env $e timeout -k 10 300 python -u tools/bench_configs.py --only $cfg --sweep-frames 6 --frames 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  %-30s %8.3f ms kernel %8.3f' % (d['config'], d['ms_per_frame'], d['kernel_ms_per_frame']))" || exit 1; done
