#!/bin/bash
# time the F(4x4) kernel variants under gpurun_variants/ (GPU box): plain / BN-prologue forward on the big SNGAN launches and
# the up-sampled-input mode;  usage: tools/variants_r4.sh <variant> ...   ("base" = the in-tree library)
mkdir -p gpurun_out/r4v
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = base ]; then L=""; else L="DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_$v.so"; fi
  env $L W4_SHORT=1 timeout 300 python tools/wino4_time.py 2>&1 | grep -v amdgpu.ids | grep -E "plain|res_up" | sed 's/| F(2x2).*| F(4x4)/| F(4x4)/'
  env $L timeout 300 python tools/wino4_upin_time.py 2>&1 | grep -v amdgpu.ids | sed 's/| vs f64.*| fused /| fused /' | head -4
done | tee gpurun_out/r4v/variants_$(date +%H%M%S).txt
