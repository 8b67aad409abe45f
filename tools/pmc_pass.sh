#!/bin/bash
# rocprofv3 counter passes over a short bench run; each pass is its own run with its own time limit.
# usage: tools/pmc_pass.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...]
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
n=0
for ctrs in "$@"; do
  n=$((n+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag/p$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-download > gpurun_out/pmc_$tag/p$n.out 2> gpurun_out/pmc_$tag/p$n.err || { echo "pass $n failed"; tail -5 gpurun_out/pmc_$tag/p$n.err; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/pmc_$tag
