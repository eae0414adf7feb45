set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "local_corr" 2>&1 | tail -3
for nb in 3 5; do ROMA_LC_BANDS=$nb timeout -k 10 500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "local_corr" 2>&1 | tail -2; done
for nb in 1 2 3 4 5 6 8; do echo "== bands $nb"; ROMA_LC_BANDS=$nb timeout -k 10 200 python tools/lc_bench.py --pairs 1 --variants rows8 2>&1 | grep rows8; ROMA_LC_BANDS=$nb timeout -k 10 200 python tools/lc_pipeline_flows.py 2>&1 | grep auto; done
echo "== default"; timeout -k 10 200 python tools/lc_bench.py --pairs 1 16 --variants rows8 2>&1 | grep rows8; timeout -k 10 200 python tools/lc_pipeline_flows.py 2>&1 | grep auto
