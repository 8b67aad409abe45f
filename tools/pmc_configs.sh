#!/bin/bash
# rocprofv3 counter passes over single configurations of tools/bench_configs.py (each pass its own run
# under its own time limit), then one summary per configuration.
# usage: tools/pmc_configs.sh <tag> <config> [<config> ...]      configs: glass teapot heightfield primitives ...
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
O=gpurun_out/$tag
mkdir -p $O
FRAMES=2; WARMF=1   # 3 frames per run
for cfg in "$@"; do
  D=$O/pmc_$cfg
  rm -rf $D && mkdir -p $D
  pass() {
    timeout -s KILL 170 rocprofv3 --pmc $2 --output-format csv -d $D/$1 -- python3 tools/bench_configs.py --only $cfg --frames $FRAMES --sweep-frames $FRAMES --warm $WARMF > $D/$1.out 2> $D/$1.err || { echo "$cfg: pass $1 failed"; tail -5 $D/$1.err; return 1; }
  }
  pass fetch "FETCH_SIZE" && pass write "WRITE_SIZE" && \
  pass l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" && \
  pass l1 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" && \
  pass f64 "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" && \
  pass wait "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" && \
  pass tatd "TA_TA_BUSY_sum TD_TD_BUSY_sum" || exit 1
  python3 tools/pmc_config_summary.py $D $((FRAMES + WARMF)) $cfg > $O/pmc_${cfg}_summary.txt || exit 1
  rm -rf $D/fetch $D/write $D/l2 $D/l1 $D/f64 $D/wait $D/tatd   # raw CSVs stay on the box
  cat $O/pmc_${cfg}_summary.txt
done
