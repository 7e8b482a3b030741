#!/bin/bash
# timing probe (GPU box): the fp32 F(4x4) kernel with its weight units fetched by ordinary global loads instead of LDS-DMA
# (X3_ABL=512; +256: also without the ds_read of the weight fragment) against the in-tree build and "no DMA" (4) / "no DMA, no read" (260)
for v in base x3abl512 x3abl768 x3abl4 x3abl260 base; do
  if [ "$v" = base ]; then L=""; else L="DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_$v.so"; fi
  echo "fp32 kernel, $v: $(env $L timeout 300 python tools/wino4x_time.py --fp32 2>&1 | grep -v amdgpu.ids | tail -1)"
done | tee gpurun_out/x3/uload_probe_$(date +%H%M%S).txt
