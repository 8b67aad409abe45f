# usage: tools/ab_libs.sh "<configs>" lib1.so lib2.so ...   (paths relative to ray-tracer_amd/lib)
cfg=$1; shift
for lib in "$@"; do echo "== $lib"; RT_AMD_LIB=$PWD/ray-tracer_amd/lib/$lib timeout -k 10 300 python -u tools/bench_configs.py --only $cfg --sweep-frames 6 --frames 20 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  %-30s %8.3f ms kernel %8.3f' % (d['config'], d['ms_per_frame'], d['kernel_ms_per_frame']))" || exit 1; done
