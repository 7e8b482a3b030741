#!/bin/bash
timeout 900 python -m pytest tests/test_dcgan_gpu.py tests/test_eltwise_gpu.py tests/test_graph_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -3
for i in 1 2; do
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dcgan eager', d['value'], d['ms_per_step'])"
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline --graph 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dcgan graph', d['value'], d['ms_per_step'])"
done
