#!/bin/bash
# compile-time ablation of the fp32 F(4x4) K loop (X3_ABL bits, conv_wino4.hip): launch times with parts of the loop removed (GPU box)
mkdir -p gpurun_out/x3
for v in base x3abl1 x3abl4 x3abl260 x3abl32 x3abl36 x3abl16 x3abl48 x3abl64 base; do
  if [ "$v" = base ]; then L=""; else L="DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_$v.so"; fi
  echo "fp32 kernel, $v: $(env $L timeout 300 python tools/wino4x_time.py --fp32 2>&1 | grep -v amdgpu.ids | tail -1)"
done | tee gpurun_out/x3/fp32_ablate_$(date +%H%M%S).txt
