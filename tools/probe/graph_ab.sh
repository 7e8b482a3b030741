#!/bin/bash
# eager launches against hipGraph replay of the same global step (single GPU), SNGAN-32 / SNGAN-64, phase 1 and 2
for i in 1 2; do
for wl in sngan32 sngan64; do
for g in "" "--graph"; do
DIAGAN_QUIET=1 python bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer $g 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl [$g]', d['value'], d['ms_per_step'], d.get('host'))"
done
done
done
