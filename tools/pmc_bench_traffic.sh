#!/bin/bash
# HBM traffic of the in-pipeline local_corr launches: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py, then
# tools/pmc_traffic.py.  Writes gpurun_out/local_corr_traffic.json (copy it to profiles/ — bench.py reports it as roofline.traffic
# while its SHA-1 stamp matches the local_corr sources), then runs the default bench with that file in place.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/traffic
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --no-cpu --no-microbench --steps 3 --warmup 1"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/bench_fetch -- python3 $BENCH > $O/fetch.log 2>&1
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/bench_write -- python3 $BENCH > $O/write.log 2>&1
python3 $R/tools/pmc_traffic.py $O/bench_fetch $O/bench_write 44550979.2 > $R/gpurun_out/local_corr_traffic.json
cp $R/gpurun_out/local_corr_traffic.json $R/profiles/local_corr_traffic.json
cd $R
python3 bench.py > $R/gpurun_out/bench_line_final.json 2> $R/gpurun_out/bench_line_final.err
