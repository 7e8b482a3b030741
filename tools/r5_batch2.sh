#!/bin/bash
# round 5, GPU-box batch 2: batched weight gradients -- tests, A/B of the bench line, SNGAN-64, DCGAN
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_sngan_gpu.py tests/test_e2e_gpu.py tests/test_wgrad_finish_gpu.py tests/test_dcgan_gpu.py -x -q -k "batched or deep_copy or logit or float64 or wgrad or train_step or dcgan" 2>&1 | tail -6 | tee gpurun_out/r5/batch2_tests.txt
for rep in 1 2; do
for sw in 1 0; do
  echo "DIAGAN_WGRAD_BATCH=$sw"
  DIAGAN_WGRAD_BATCH=$sw timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sngan32', d['value'], d['ms_per_step'])"
  DIAGAN_WGRAD_BATCH=$sw timeout 300 python bench.py --workload sngan64 --steps 20 --warmup 4 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  sngan64', d['value'], d['ms_per_step'])"
  DIAGAN_WGRAD_BATCH=$sw timeout 300 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  dcgan', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r5/wgrad_batch_ab.txt
