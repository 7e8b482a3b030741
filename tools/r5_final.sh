#!/bin/bash
# round-5 final measurements on the GPU box (through gpurun): profiles (kernel trace + PMC passes), the default bench line,
# the other workloads' lines, the StyleGAN2 line.  Output under gpurun_out/ (copied into profiles/ afterwards).
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh r05 > /dev/null 2>&1
mkdir -p $R/gpurun_out/r5raw
cd $R
python bench.py > gpurun_out/r5raw/bench.json 2> gpurun_out/r5raw/bench.err
python bench.py --workload sngan64 --no_cpu_baseline 2>/dev/null | grep "^{" > gpurun_out/r5raw/bench64.json
python bench.py --phase 2 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | grep "^{" > gpurun_out/r5raw/bench_p2.json
python bench.py --workload sngan64 --phase 2 --no_cpu_baseline 2>/dev/null | grep "^{" > gpurun_out/r5raw/bench64_p2.json
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null | grep "^{" > gpurun_out/r5raw/bench_dcgan.json
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline --graph 2>/dev/null | grep "^{" > gpurun_out/r5raw/bench_dcgan_graph.json
python bench.py --workload stylegan2 --steps 8 --warmup 3 2>/dev/null | grep "^{" > gpurun_out/r5raw/bench_sg2.json
ls -la $R/gpurun_out/r5raw
for f in bench bench64 bench_p2 bench64_p2 bench_dcgan bench_dcgan_graph bench_sg2; do python -c "import json,sys; d=json.loads(open('gpurun_out/r5raw/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
