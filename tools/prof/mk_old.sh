#!/bin/bash
# tools/prof/mk_old.sh a.hip [b.hip ...]: tools/scratch/libroma_hip_old.so = the in-tree objects with the named sources taken from HEAD (A/B on one box: ab.sh)
set -e
cd /root/repo/roma_amd/csrc
objs=""
for o in local_corr local_corr_t8 local_corr_rows sampling cls_refine cos_kernel chol finalize kde dwconv pointwise refiner_head refiner_block refiner_wide bias_relu preproc tiny_corr layernorm sample attention jpeg error; do
  use=$o.o
  for f in "$@"; do
    if [ "$f" = "$o.hip" ]; then
      git show HEAD:roma_amd/csrc/$f > old_tmp_$f
      extra=""; [ $o = local_corr_rows ] && extra="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"
      /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 $extra -c old_tmp_$f -o /tmp/old_$o.o
      rm old_tmp_$f
      use=/tmp/old_$o.o
    fi
  done
  objs="$objs $use"
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/scratch/libroma_hip_old.so $objs
