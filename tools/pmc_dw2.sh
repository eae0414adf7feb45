#!/bin/bash
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2pmc_dw2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1 ctr=$2; shift 2; echo "[$(date +%T)] pass $name: $ctr" | tee -a $O/progress.txt
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/$name -- python3 "$@" > $O/$name.log 2>&1; }
for cfg in "d576h216 576 216" "d144h432 144 432"; do
  set -- $cfg; tag=$1; shift
  DW="$R/tools/dw_micro.py $*"
  pass ${tag}_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" $DW
  pass ${tag}_sq2 "SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $DW
  pass ${tag}_fetch "FETCH_SIZE" $DW
  pass ${tag}_tcc "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" $DW
done
python3 $R/tools/pmc_table.py $O dwconv > $O/summary.txt
