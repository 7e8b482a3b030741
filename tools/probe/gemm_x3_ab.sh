#!/bin/bash
# the lone-tile implicit-GEMM launches on the bf16 pipe with split operands (DIAGAN_GEMM_X3=1) against the fp32 kernel (=0)
timeout 900 python -m pytest tests/test_conv_gpu.py -x -q -k "split_operand" 2>&1 | tail -12
for i in 1 2; do
for b in 0 1; do
for wl in sngan32 sngan64; do
DIAGAN_GEMM_X3=$b DIAGAN_QUIET=1 python bench.py --workload $wl --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemm_x3 $b $wl', d['value'], d['ms_per_step'])"
done
done
done
