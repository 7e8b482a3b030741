#!/bin/bash
timeout 900 python -m pytest tests/test_eltwise_gpu.py tests/test_sngan_gpu.py -x -q -k "grouped or batchnorm or stacked or train_steps or full_batch" 2>&1 | tail -4
for i in 1 2; do
for wl in sngan32 sngan64; do
DIAGAN_QUIET=1 python bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', d['value'], d['ms_per_step'])"
done
done
