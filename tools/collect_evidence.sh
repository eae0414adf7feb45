#!/bin/bash
# Round evidence on the MI355X box (gpurun): kernel trace + stats of the bench command, last-step table, PMC of the depthwise
# kernel, stand-alone kernel timings.  Outputs under gpurun_out/evidence/; copy what is to be judged into profiles/.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "[$(date +%T)] kernel trace of bench.py" | tee -a $O/progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu > $O/bench_prof.json 2> $O/bench_prof.err
python3 $R/tools/trace_last_step.py $(ls $O/prof_bench/*/*kernel_trace.csv | head -1) > $O/bench_last_step.md
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
pass() { local name=$1 ctr=$2; shift 2; echo "[$(date +%T)] pass $name: $ctr" | tee -a $O/progress.txt
  timeout -k 5 90 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_dw/$name -- python3 "$@" > $O/$name.log 2>&1; }   # a rejected counter set hangs the child: never without a timeout
for cfg in "d576h216 576 216" "d144h432 144 432" "d1144h108 1144 108"; do
  set -- $cfg; tag=$1; shift
  DW="$R/tools/dw_micro.py $*"
  pass ${tag}_sq1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" $DW
  pass ${tag}_sq2 "SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" $DW
  pass ${tag}_fetch "FETCH_SIZE" $DW
  pass ${tag}_write "WRITE_SIZE" $DW
  pass ${tag}_tcc "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" $DW
done
python3 $R/tools/pmc_table.py $O/pmc_dw dwconv > $O/dwconv_pmc_summary.txt
rm -rf $O/prof_bench/*/*_agent_info.csv
cd $R
echo "[$(date +%T)] kernel microbench" | tee -a $O/progress.txt
python3 tools/bench_kernels.py --pairs 1 > $O/kernels_microbench.txt 2>&1
python3 tools/bench_kernels.py --pairs 1 --flow adversarial 2>&1 | grep local_corr >> $O/kernels_microbench.txt
python3 tools/bench_kernels.py --pairs 16 2>&1 | grep -E "local_corr|dwconv" >> $O/kernels_microbench.txt
echo "[$(date +%T)] bench lines" | tee -a $O/progress.txt
python3 bench.py > $O/bench_line.json 2> $O/bench_line.err
python3 bench.py --pairs 8 --no-cpu --no-microbench > $O/bench_pairs8.json 2>> $O/bench_line.err
python3 bench.py --graph --no-cpu --no-microbench > $O/bench_graph.json 2>> $O/bench_line.err
python3 bench.py --workload indoor_sample --no-cpu --no-microbench > $O/bench_indoor_sample.json 2>> $O/bench_line.err
python3 bench.py --workload tiny --no-cpu --no-microbench > $O/bench_tiny.json 2>> $O/bench_line.err
echo "[$(date +%T)] done" | tee -a $O/progress.txt
