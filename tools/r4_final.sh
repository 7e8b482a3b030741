#!/bin/bash
# round-4 final measurements on the GPU box (through gpurun): profiles (kernel trace + PMC passes), the default bench line,
# the other workloads' lines, the StyleGAN2 kernel trace.  Output under gpurun_out/ (copied into profiles/ afterwards).
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh r04 > /dev/null 2>&1
mkdir -p $R/gpurun_out/r4raw
cd $R
python bench.py > gpurun_out/r4raw/bench.json 2> gpurun_out/r4raw/bench.err
python bench.py --workload sngan64 --no_cpu_baseline 2>/dev/null | grep "^{" > gpurun_out/r4raw/bench64.json
python bench.py --phase 2 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | grep "^{" > gpurun_out/r4raw/bench_p2.json
python bench.py --workload sngan64 --phase 2 --no_cpu_baseline 2>/dev/null | grep "^{" > gpurun_out/r4raw/bench64_p2.json
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null | grep "^{" > gpurun_out/r4raw/bench_dcgan.json
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline --graph 2>/dev/null | grep "^{" > gpurun_out/r4raw/bench_dcgan_graph.json
python bench.py --workload stylegan2 --steps 8 --warmup 3 2>/dev/null | grep "^{" > gpurun_out/r4raw/bench_sg2.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04/kt_sg2 -- python3 $R/bench.py --workload stylegan2 --steps 6 --warmup 2 --no_cpu_baseline > $R/gpurun_out/prof_r04/kt_sg2.log 2>&1
find $R/gpurun_out/prof_r04/kt_sg2 -name "*kernel_trace.csv" -delete
find $R/gpurun_out/prof_r04 -name "*.db" -delete
python $R/tools/sngan_f64_parity.py > $R/gpurun_out/r4raw/f64_parity.txt 2>&1
ls $R/gpurun_out/r4raw
