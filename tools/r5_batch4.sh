#!/bin/bash
# round 5, GPU-box batch 4: implicit-GEMM weight gradients batched too -- tests and A/B
mkdir -p gpurun_out/r5
timeout 1200 python -m pytest tests/test_sngan_gpu.py tests/test_dcgan_gpu.py tests/test_dp_gpu.py tests/test_graph_gpu.py tests/test_e2e_gpu.py -x -q -k "not trajectory and not float64" 2>&1 | tail -6 | tee gpurun_out/r5/batch4_tests.txt
for rep in 1 2; do
for sw in 1 0; do
  echo "DIAGAN_WGRAD_BATCH=$sw"
  DIAGAN_WGRAD_BATCH=$sw timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sngan32', d['value'], d['ms_per_step'])"
  DIAGAN_WGRAD_BATCH=$sw timeout 300 python bench.py --workload sngan64 --steps 20 --warmup 4 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sngan64', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r5/wgrad_batch_ab2.txt
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dcgan graph', d['value'], d['ms_per_step'])"
python bench.py --phase 2 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sngan32 phase 2', d['value'], d['ms_per_step'])"
