#!/bin/bash
for i in 1 2 3; do
python bench.py --phase 2 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p2 default', d['value'], d['ms_per_step'], d['host']['launch_loop_ms_per_step'])"
python bench.py --phase 2 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p2 no timer', d['value'], d['ms_per_step'], d['host']['launch_loop_ms_per_step'])"
DIAGAN_WGRAD_BATCH=0 python bench.py --phase 2 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('p2 no batch', d['value'], d['ms_per_step'], d['host']['launch_loop_ms_per_step'])"
done
