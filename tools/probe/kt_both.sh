#!/bin/bash
# kernel-trace statistics of one SNGAN-32 and one SNGAN-64 run (GPU box): the non-GEMM list after a change
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r05k
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt32 -- python3 $R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer > $OUT/kt32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt64 -- python3 $R/bench.py --workload sngan64 --steps 5 --warmup 2 --no_cpu_baseline --no_kernel_timer > $OUT/kt64.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.db" -delete
ls $OUT/kt32/*/ $OUT/kt64/*/
