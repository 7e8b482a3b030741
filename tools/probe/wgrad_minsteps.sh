#!/bin/bash
# which layers should share a batched weight-gradient launch: threshold sweep (GPU box)
for rep in 1 2; do for ms in 12 20 24 40 100000; do
  DIAGAN_WGRAD_BATCH_MIN_STEPS=$ms timeout 300 python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_steps $ms sngan32', d['value'], d['ms_per_step'])"
  DIAGAN_WGRAD_BATCH_MIN_STEPS=$ms timeout 300 python bench.py --workload sngan64 --steps 20 --warmup 4 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_steps $ms sngan64', d['value'], d['ms_per_step'])"
done; done
