#!/bin/bash
# split-K combine by the last-arriving workgroup (DIAGAN_SPLITK_FUSED=1, default) against the second launch (=0): same box
timeout 900 python -m pytest tests/test_wino_gpu.py -x -q 2>&1 | tail -8
for i in 1 2; do
for b in 0 1; do
for wl in sngan32 sngan64; do
DIAGAN_SPLITK_FUSED=$b DIAGAN_QUIET=1 python bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused $b $wl', d['value'], d['ms_per_step'])"
done
done
done
