#!/bin/bash
# Round-2 PMC collection for roma_local_corr (run on the MI355X box through gpurun; every pass is its own rocprofv3 run,
# --pmc beside --kernel-trace only).  Writes gpurun_out/r2pmc/<pass>/ and gpurun_out/r2pmc/summary.txt.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() {  # pass <name> "<counters>" <program args...>
  local name=$1 ctr=$2; shift 2
  echo "[$(date +%T)] pass $name: $ctr" | tee -a $O/progress.txt
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -- python3 "$@" > $O/$name.log 2>&1
}
BENCH="$R/bench.py --no-cpu --no-microbench --steps 3 --warmup 1"
pass bench_fetch "FETCH_SIZE" $BENCH
pass bench_write "WRITE_SIZE" $BENCH
python3 $R/tools/pmc_traffic.py $O/bench_fetch $O/bench_write 44550979.2 > $O/local_corr_traffic.json
for cfg in "U4p8_auto U4 --pairs 8" "U4p8_t84 U4 --pairs 8 --variant tile8x4" "U4p1_auto U4 --pairs 1"; do
  set -- $cfg; tag=$1; shift
  LC="$R/tools/lc_micro.py $* --iters 3"
  pass ${tag}_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU" $LC
  pass ${tag}_fetch "FETCH_SIZE" $LC
  pass ${tag}_write "WRITE_SIZE" $LC
  pass ${tag}_tcc "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" $LC
  pass ${tag}_sq2 "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" $LC
done
python3 $R/tools/pmc_table.py $O > $O/summary.txt
# last: counters whose names this ROCm may not know (a refusal here loses nothing above)
pass U4p8_auto_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" $R/tools/lc_micro.py U4 --pairs 8 --iters 3
python3 $R/tools/pmc_table.py $O > $O/summary.txt
