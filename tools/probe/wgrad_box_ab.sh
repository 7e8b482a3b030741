#!/bin/bash
# box sums in the pooled weight gradient's loader (DIAGAN_WGRAD_BOX=1, default) against the boxsum2 pass in front (=0): same box
timeout 900 python -m pytest tests/test_wino_gpu.py tests/test_sngan_gpu.py -x -q -k "pool or box or batched" 2>&1 | tail -3
for i in 1 2; do
for b in 0 1; do
for wl in sngan32 sngan64; do
DIAGAN_WGRAD_BOX=$b DIAGAN_QUIET=1 python bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('box $b $wl', d['value'], d['ms_per_step'])"
done
done
done
