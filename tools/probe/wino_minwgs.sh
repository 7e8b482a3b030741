#!/bin/bash
# does the F(2x2) kernel beat the implicit GEMM on the 8x8 maps if its workgroup floor (192) is lowered?  (timing only)
for w in 192 128 64 32; do
for wl in sngan32 sngan64; do
DIAGAN_WINO_MIN_WGS=$w DIAGAN_QUIET=1 python bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_wgs $w $wl', d['value'], d['ms_per_step'])"
done
done
