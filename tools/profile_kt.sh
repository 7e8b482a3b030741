#!/bin/bash
# kernel-trace stats only (no PMC passes) of the SNGAN-32 and SNGAN-64 steps: gpurun_out/prof_<tag>/kt{32,64}
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt32 -- python3 $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_sngan64_leg > $OUT/kt32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt64 -- python3 $R/bench.py --workload sngan64 --steps 5 --warmup 2 --no_cpu_baseline > $OUT/kt64.log 2>&1
find $OUT/kt32 $OUT/kt64 -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
