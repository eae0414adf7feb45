# A/B of a micro-benchmark command on one box: bash tools/prof/ab_micro.sh "python tools/rb_micro.py 24 32 864"
cd $GRAFT_REPO_ROOT
cp roma_amd/csrc/libroma_hip.so /tmp/new.so
for rep in 1 2; do
  for v in old new; do
    if [ $v = old ]; then cp tools/scratch/libroma_hip_old.so roma_amd/csrc/libroma_hip.so; else cp /tmp/new.so roma_amd/csrc/libroma_hip.so; fi
    echo "== $v"; eval "$1" 2>&1 | tail -${2:-3}
  done
done
cp /tmp/new.so roma_amd/csrc/libroma_hip.so
