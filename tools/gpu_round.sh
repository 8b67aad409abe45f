#!/bin/bash
# One GPU session: smoke, bench, rocprofv3 kernel trace, HBM-traffic counter passes.  Every GPU step
# runs under its own time limit and the steps are chained, so trouble ends the script.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1 || { echo "smoke failed"; tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 300 python -u bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err || { echo "bench failed"; tail -20 gpurun_out/bench.err; exit 1; }
cat gpurun_out/bench.json
rm -rf gpurun_out/prof gpurun_out/pmc_traffic
# the headline leg alone (frames stay in HBM): every k_generation launch in the table belongs to the loop whose
# launches bench.py times with HIP events, so the two averages are comparable
rm -rf gpurun_out/prof_nocopy
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nocopy -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-download > gpurun_out/prof_nocopy_bench.json 2> gpurun_out/prof_nocopy.err || { echo "rocprof (no copy) failed"; tail -20 gpurun_out/prof_nocopy.err; exit 1; }
find gpurun_out/prof_nocopy -name "*kernel_stats*.csv" | head -3
# both legs; rocprofv3 turns the host copies of the second leg into copy kernels that stretch the render kernel
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.json 2> gpurun_out/prof.err || { echo "rocprof failed"; tail -20 gpurun_out/prof.err; exit 1; }
mkdir -p gpurun_out/pmc_traffic
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_traffic/p1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-download > gpurun_out/pmc_traffic/p1.out 2> gpurun_out/pmc_traffic/p1.err || { echo "FETCH_SIZE pass failed"; tail -5 gpurun_out/pmc_traffic/p1.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_traffic/p2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-download > gpurun_out/pmc_traffic/p2.out 2> gpurun_out/pmc_traffic/p2.err || { echo "WRITE_SIZE pass failed"; tail -5 gpurun_out/pmc_traffic/p2.err; exit 1; }
python3 tools/traffic_json.py gpurun_out/pmc_traffic gpurun_out/traffic.json
