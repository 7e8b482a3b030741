#!/bin/bash
# kernel-trace statistics of the StyleGAN2 256 x 256 iteration (GPU box): gpurun_out/prof_<tag>/sg2 + a markdown summary
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sg2 -- python3 $R/bench.py --workload stylegan2 --steps 6 --warmup 2 --no_cpu_baseline > $OUT/sg2.log 2>&1
find $OUT/sg2 -name "*kernel_trace.csv" -delete
find $OUT -name "*.db" -delete
cd $R
python tools/prof_summary.py $(find $OUT/sg2 -name "*kernel_stats.csv" | head -1) 10 $OUT/sg2_summary.md "StyleGAN2 256x256 batch 32 ($TAG)" > /dev/null
tail -2 $OUT/sg2.log | head -c 400
