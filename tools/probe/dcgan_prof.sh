#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r05q
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
DIAGAN_QUIET=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktdc -- python3 $R/bench.py --workload dcgan --steps 10 --warmup 3 --no_cpu_baseline --no_kernel_timer > $OUT/ktdc.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
tail -2 $OUT/ktdc.log | cut -c1-300
