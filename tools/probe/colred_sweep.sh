#!/bin/bash
# column reductions (BatchNorm statistics / backward sums, bias gradients): rows per split and block count
for cfg in "256 1024" "128 2048" "64 4096" "32 4096"; do
set -- $cfg
for wl in sngan32 sngan64; do
DIAGAN_COLRED_ROWS=$1 DIAGAN_COLRED_BLOCKS=$2 DIAGAN_QUIET=1 python bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows $1 blocks $2 $wl', d['value'], d['ms_per_step'])"
done
done
