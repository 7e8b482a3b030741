#!/bin/bash
# ULOAD in the pooled F(4x4) modes (one instantiation per K-loop remainder mod 3) against their LDS-DMA build (GPU box)
timeout 900 python -m pytest tests/test_wino4_gpu.py -x -q -k "pool" 2>&1 | tail -3
for rep in 1 2; do for v in base pooldma; do
  if [ "$v" = base ]; then L=""; else L="DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_$v.so"; fi
  env $L timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['roofline']['all_gemm_kernels_2_untimed_steps']
print('$v sngan32', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in t.items() if ',1,false>' in k or ',2,false>' in k})"
  env $L timeout 300 python bench.py --workload sngan64 --steps 20 --warmup 4 --no_cpu_baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['roofline']['all_gemm_kernels_2_untimed_steps']
print('$v sngan64', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in t.items() if ',1,false>' in k or ',2,false>' in k})"
done; done
