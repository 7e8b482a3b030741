#!/bin/bash
# BatchNorm rows from LDS (ULOAD modes) against two vector-memory loads per K-step: tests, launch times, bench (GPU box)
timeout 900 python -m pytest tests/test_wino4_gpu.py tests/test_sngan_gpu.py -x -q -k "not trajectory and not float64" 2>&1 | tail -3
for v in base noprolds base noprolds; do
  if [ "$v" = base ]; then L=""; else L="DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_$v.so"; fi
  echo "$v: $(env $L timeout 300 python tools/wino4x_time.py --fp32 2>&1 | grep -v amdgpu.ids | tail -1)"
done
for rep in 1 2; do for v in base noprolds; do
  if [ "$v" = base ]; then L=""; else L="DIAGAN_LIB_PATH=$PWD/gpurun_variants/libdiagan_$v.so"; fi
  env $L timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v sngan32', d['value'], d['ms_per_step'])"
  env $L timeout 300 python bench.py --workload sngan64 --steps 20 --warmup 4 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v sngan64', d['value'], d['ms_per_step'])"
done; done
