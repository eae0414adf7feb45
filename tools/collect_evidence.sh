#!/bin/bash
# Round evidence on the MI355X box (gpurun): kernel trace + stats of the bench command, last-step table, PMC of the local_corr
# row-streaming kernel and of the fused wide refiner block, stand-alone kernel timings, HBM traffic of the in-pipeline local_corr
# launches, the bench lines.  Outputs under gpurun_out/evidence/; copy what is to be judged into profiles/.
# Every rocprofv3 run is wrapped in `timeout` (a rejected counter set leaves the profiled child hung) and has the program directly
# behind `--`.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
PART=${1:-a}            # a: trace + counters + micro-benchmarks; b: traffic + bench lines (two gpurun calls: each stays under 20 min)
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
say() { echo "[$(date +%T)] $*" | tee -a $O/progress_$PART.txt; }
if [ "$PART" = "a" ]; then
say "kernel trace of bench.py"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu --no-microbench > $O/bench_prof.json 2> $O/bench_prof.err
python3 $R/tools/trace_last_step.py $(ls $O/prof_bench/*/*kernel_trace.csv | head -1) > $O/bench_last_step.md
cp $(ls $O/prof_bench/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
rm -f $O/prof_bench/*/*kernel_trace.csv $O/prof_bench/*/*agent_info.csv
say "PMC: local_corr rows kernel, U4 at 8 pairs, coherent"
bash $R/tools/pmc_kernel.sh evidence/pmc_lc_u4 local_corr tools/lc_micro.py U4 --pairs 8 > $O/pmc_lc_u4.txt 2>&1
say "PMC: local_corr rows kernel, U8 at 8 pairs, coherent"
bash $R/tools/pmc_kernel.sh evidence/pmc_lc_u8 local_corr tools/lc_micro.py U8 --pairs 8 > $O/pmc_lc_u8.txt 2>&1
say "PMC: refiner_wide at 216^2"
bash $R/tools/pmc_kernel.sh evidence/pmc_rw refiner_wide tools/rw_micro.py 216 > $O/pmc_rw.txt 2>&1
find $O -name "*agent_info.csv" -delete
cd $R
say "kernel microbenchmarks"
python3 tools/lc_bench.py > $O/lc_bench_coherent.txt 2>&1
python3 tools/lc_bench.py --flow adversarial --pairs 1 8 > $O/lc_bench_adversarial.txt 2>&1
python3 tools/lc_pipeline_flows.py > $O/lc_pipeline_flows.txt 2>&1
python3 tools/rw_micro.py > $O/rw_micro.txt 2>&1
python3 tools/bench_kernels.py --pairs 1 > $O/kernels_microbench.txt 2>&1
tools/scratch/stage_micro 32 > $O/stage_micro.txt 2>&1
say "done (a)"
exit 0
fi
cd $R
say "HBM traffic of the in-pipeline local_corr launches + the default bench line"
bash tools/pmc_bench_traffic.sh > $O/traffic.log 2>&1
cp gpurun_out/bench_line_final.json $O/bench_line.json 2>/dev/null
cp gpurun_out/local_corr_traffic.json $O/ 2>/dev/null
say "other bench lines"
python3 bench.py --pairs 8 --no-cpu --no-microbench > $O/bench_pairs8.json 2>> $O/bench_line.err
python3 bench.py --workload coarse --no-cpu --no-microbench > $O/bench_coarse.json 2>> $O/bench_line.err
python3 bench.py --workload indoor_sample --no-cpu --no-microbench > $O/bench_indoor_sample.json 2>> $O/bench_line.err
python3 bench.py --workload tiny --no-cpu --no-microbench > $O/bench_tiny.json 2>> $O/bench_line.err
say "done (b)"
