#!/bin/bash
# the pooled shortcut's gradient un-pooled by c1's data-gradient epilogue (DIAGAN_RES_UNPOOL=1, default) against avgpool2_bwd (=0)
timeout 900 python -m pytest tests/test_wino_gpu.py tests/test_sngan_gpu.py -x -q -k "unpooled or full_batch or train_steps" 2>&1 | tail -5
for i in 1 2; do
for b in 0 1; do
for wl in sngan32 sngan64; do
DIAGAN_RES_UNPOOL=$b DIAGAN_QUIET=1 python bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('res_unpool $b $wl', d['value'], d['ms_per_step'])"
done
done
done
