#!/bin/bash
# One GPU session: smoke, bench, rocprofv3 kernel trace (stats + timeline), counter passes.  Every GPU
# step runs under its own time limit and the steps are chained, so trouble ends the script.
# usage: tools/gpu_round.sh <tag>      results under gpurun_out/<tag>/
set -o pipefail
tag=${1:-round}
O=gpurun_out/$tag
rm -rf $O && mkdir -p $O
export TMPDIR=/tmp
# 20 timed frames: a ring of 8, two frames per launch and one at each end of a stretch (FrameLoop::run), so 17 priming
# and 4 warm-up launches precede the 11 timed ones: launches 21..31
B="--steps 20 --warmup 5"
timeout -k 10 200 python -u -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python -u bench.py $B > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -20 $O/bench.err; exit 1; }
# the loop that leaves the frames in HBM, alone: under rocprofv3 the runtime turns the DMA copies of the metric's
# loop into copy kernels that share the GPU with the render kernel, so that loop is traced separately below
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_hbm -- python3 bench.py $B --no-cpu-baseline --hbm-only > $O/trace_hbm_bench.json 2> $O/trace_hbm.err || { echo "rocprof (hbm) failed"; tail -20 $O/trace_hbm.err; exit 1; }
python3 tools/timeline.py $O/trace_hbm 21 11 > $O/timeline_hbm.txt || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_host -- python3 bench.py $B --no-cpu-baseline --no-hbm-leg > $O/trace_host_bench.json 2> $O/trace_host.err || { echo "rocprof (host) failed"; tail -20 $O/trace_host.err; exit 1; }
python3 tools/timeline.py $O/trace_host 21 11 > $O/timeline_host.txt || exit 1
P="--steps 3 --warmup 1 --no-cpu-baseline --hbm-only"
pass() {  # name, counters; one frame per launch, so that "per launch" is per frame
  RT_AMD_FRAMES_PER_LAUNCH=1 timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d $O/pmc/$1 -- python3 bench.py $P > $O/pmc_$1.out 2> $O/pmc_$1.err || { echo "pmc pass $1 failed"; tail -5 $O/pmc_$1.err; exit 1; }
}
mkdir -p $O/pmc
pass fetch "FETCH_SIZE" && pass write "WRITE_SIZE" && \
pass f64 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" && \
pass wait "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE" && \
pass cache "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" && \
pass tatd "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" || exit 1
mkdir -p $O/pmc_traffic && cp -r $O/pmc/fetch $O/pmc_traffic/p1 && cp -r $O/pmc/write $O/pmc_traffic/p2
python3 tools/traffic_json.py $O/pmc_traffic $O/traffic.json > /dev/null || exit 1
python3 tools/pmc_f64_json.py $O/pmc $O/pmc_f64.json > /dev/null || exit 1
python3 tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt || exit 1
for t in trace_hbm trace_host; do f=$(find $O/$t -name "*kernel_stats.csv" -print -quit); [ -n "$f" ] && cp "$f" $O/${t}_kernel_stats.csv; done
# keep the merged output small: the raw traces and counter CSVs stay on the box
rm -rf $O/trace_hbm $O/trace_host $O/pmc $O/pmc_traffic
ls $O
# what one GPU of an N-GPU run does (part 0..N-1 of N alone on this GPU), copy rates, host registration
timeout -k 10 300 python3 tools/part_period.py 0 8 > $O/part_period.txt 2>&1 || { echo "part_period failed"; exit 1; }
# the same with one frame per launch and copy in a ring of eight (what the loop did before it launched several at once)
RT_AMD_FRAMES_PER_LAUNCH=1 timeout -k 10 200 python3 tools/part_period.py 8 8 > $O/part_period_one_frame_per_launch.txt 2>&1 || { echo "part_period (one frame per launch) failed"; exit 1; }
if [ -x ray-tracer_amd/lib/d2h_probe ]; then timeout -k 10 100 ray-tracer_amd/lib/d2h_probe > $O/d2h_probe.txt 2>&1 || exit 1; fi
if [ -x ray-tracer_amd/lib/register_probe ]; then timeout -k 10 100 ray-tracer_amd/lib/register_probe > $O/register_probe.txt 2>&1 || exit 1; fi
# the launcher path, two ranks on this one GPU (gloo for the barriers): a rehearsal, no scaling figure
RT_BENCH_BACKEND=gloo timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_2ranks_gloo_rehearsal.json 2> $O/bench_2ranks.err || { echo "2-rank rehearsal failed"; exit 1; }
# per-wave run times and loop iterations (diagnostic builds, when present)
if [ -f ray-tracer_amd/lib/librt_host_wclk.so ] && [ -f ray-tracer_amd/lib/librt_host_wit.so ]; then
  RT_AMD_LIB=librt_amd_wclk.so RT_HOST_LIB=librt_host_wclk.so timeout -k 10 200 python3 tools/wave_clock.py > $O/wave_clock.jsonl 2> /dev/null || exit 1
  RT_AMD_LIB=librt_amd_wit.so RT_HOST_LIB=librt_host_wit.so timeout -k 10 200 python3 tools/wave_clock.py > /dev/null 2>&1 || exit 1
  python3 tools/wave_iters.py gpurun_out/wave_wclk_teapot.npy gpurun_out/wave_wit_teapot.npy > $O/wave_iterations.txt || exit 1
fi
ls $O
