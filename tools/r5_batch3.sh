#!/bin/bash
# round 5, GPU-box batch 3: the whole GPU suite + SNGAN-64 kernel stats
mkdir -p gpurun_out/r5
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r5/gpu_suite.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r05q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt64 -- python3 $R/bench.py --workload sngan64 --steps 5 --warmup 2 --no_cpu_baseline > $OUT/kt64.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
