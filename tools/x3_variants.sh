#!/bin/bash
# time the X3 F(4x4) kernel's compile-time ablation variants under gpurun_variants/ (GPU box): tools/x3_variants.sh <variant> ...
# ("base" = the in-tree library, "fp32" = the in-tree library with X3 off)
mkdir -p gpurun_out/x3
for v in "$@"; do
  if [ "$v" = base ]; then L=""; A=""; elif [ "$v" = fp32 ]; then L=""; A="--fp32"; else L="DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_$v.so"; A=""; fi
  echo "$v: $(env $L timeout 300 python tools/wino4x_time.py $A 2>&1 | grep -v amdgpu.ids | tail -1)"
done | tee gpurun_out/x3/variants_$(date +%H%M%S).txt
