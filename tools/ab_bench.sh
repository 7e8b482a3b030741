#!/bin/bash
# A/B: bench.py with and without an environment switch (GPU box).  usage: tools/ab_bench.sh VAR [workload]
VAR=$1; WL=${2:-sngan32}
for v in 0 1 0 1; do
  echo -n "$VAR=$v: "
  env $VAR=$v python bench.py --workload $WL --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
