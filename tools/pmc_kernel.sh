#!/bin/bash
# Standard counter passes over ONE micro-program (run on the MI355X box through gpurun):
#   tools/pmc_kernel.sh <out-dir under gpurun_out/> <kernel-name substring> <python script> [args...]
# e.g. tools/pmc_kernel.sh r2pmc_dw4 dwconv tools/dw_micro.py 576 216
# Every pass is its own rocprofv3 run (--pmc beside --kernel-trace only) under its own `timeout`: on gfx950 at most two TA / TCP / TD
# counters fit one pass (four abort rocprofv3 with "exceeds the capabilities of the hardware" and leave the child hung), FETCH_SIZE
# and WRITE_SIZE need separate passes, and SQ takes eight counters.  A failed pass ends the script; the table of what was collected
# is printed at the end (tools/pmc_table.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; K=$2; shift 2
PROG="$R/$1"; shift
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1 ctr=$2; timeout -k 5 60 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -- python3 $PROG "${ARGS[@]}" > $O/$name.log 2>&1 || { echo "pass $name failed"; return 1; }; }
ARGS=("$@")
{ pass sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" &&
  pass sq2 "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR" &&
  pass fetch "FETCH_SIZE" && pass write "WRITE_SIZE" && pass tcc "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" &&
  pass ta "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES" && pass tcp1 "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY" &&
  pass tcp2 "TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES" && pass td "TD_TD_BUSY TD_TC_STALL"; } || true
python3 $R/tools/pmc_table.py $O $K | tee $O/summary.txt
