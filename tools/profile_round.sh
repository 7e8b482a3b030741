#!/bin/bash
# Round profiles on the GPU box (run through gpurun): rocprofv3 kernel-trace stats of the default bench.py run and of
# the SNGAN-64 workload, then the PMC passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes):
# FETCH_SIZE, WRITE_SIZE, and SQ_VALU_MFMA_BUSY_CYCLES + SQ_BUSY_CU_CYCLES + GRBM_GUI_ACTIVE.  Output: gpurun_out/prof_<tag>/...
# usage: tools/profile_round.sh <tag>
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B32="$R/bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_sngan64_leg"
B64="$R/bench.py --workload sngan64 --steps 5 --warmup 2 --no_cpu_baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt32 -- python3 $B32 > $OUT/kt32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt64 -- python3 $B64 > $OUT/kt64.log 2>&1
PM="--steps 3 --warmup 2 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc32_$C -- python3 $R/bench.py $PM > $OUT/pmc32_$C.log 2>&1
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc64_$C -- python3 $R/bench.py --workload sngan64 $PM > $OUT/pmc64_$C.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc32_mfma -- python3 $R/bench.py $PM > $OUT/pmc32_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc64_mfma -- python3 $R/bench.py --workload sngan64 $PM > $OUT/pmc64_mfma.log 2>&1
# keep the summaries, drop the per-dispatch traces of the stats runs (large)
find $OUT/kt32 $OUT/kt64 -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
ls -la $OUT $OUT/*/* | head -60
