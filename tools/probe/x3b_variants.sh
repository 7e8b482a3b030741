#!/bin/bash
# run tools/probe/gemm_x3b_time.py (short form) for the library variants under gpurun_variants/ named on the command line (GPU box)
mkdir -p gpurun_out/r6
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = base ]; then XB_SHORT=1 timeout 200 python tools/probe/gemm_x3b_time.py 2>&1 | grep -v amdgpu.ids
  else DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_$v.so XB_SHORT=1 timeout 200 python tools/probe/gemm_x3b_time.py 2>&1 | grep -v amdgpu.ids; fi
done | tee gpurun_out/r6/x3b_variants_$(date +%H%M%S).txt
