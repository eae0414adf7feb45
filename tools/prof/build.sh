#!/bin/bash
# tools/prof/build.sh chol_prof|rw_prof|toep_prof|valu_rate: generate the instrumented copy of the kernel source (gen.py) and build it into
# tools/scratch/<name>/ (git-ignored, travels to the GPU box with the snapshot); then on the box: python tools/prof/<name>/run.py
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
n=$1
mkdir -p $R/tools/scratch/$n
if [ "$n" = valu_rate ]; then
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o $R/tools/scratch/$n/valu_rate $R/tools/prof/valu_rate/valu_rate.hip
else
  python3 $R/tools/prof/$n/gen.py
  (cd $R/tools/scratch/$n && /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -shared -o lib$n.so $n.hip ../../../roma_amd/csrc/error.cpp)
fi
