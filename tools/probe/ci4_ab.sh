#!/bin/bash
# the first-layer kernel (conv3x3_ci4) against the implicit GEMM on the same box
python -m pytest tests/test_conv_gpu.py -x -q -k "first_layer or fwd" 2>&1 | tail -3
python tools/probe/first_conv_tiles.py 2>&1 | grep "B="
for i in 1 2; do
for b in 0 1; do
for wl in sngan32 sngan64; do
DIAGAN_CONV_CI4=$b DIAGAN_QUIET=1 python bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ci4 $b $wl', d['value'], d['ms_per_step'])"
done
done
done
