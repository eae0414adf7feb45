# A/B of two builds of the library on the same box: tools/scratch/libroma_hip_old.so vs the in-tree one
cd $GRAFT_REPO_ROOT
cp roma_amd/csrc/libroma_hip.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp tools/scratch/libroma_hip_old.so roma_amd/csrc/libroma_hip.so; else cp /tmp/new.so roma_amd/csrc/libroma_hip.so; fi
    python bench.py --no-cpu --no-microbench 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],2), round(d['ms_per_step'],3))"
  done
done
cp /tmp/new.so roma_amd/csrc/libroma_hip.so
