#!/bin/bash
# round-6 final measurements on the GPU box (through gpurun): profiles (kernel trace + PMC passes of the SNGAN workloads, kernel trace of
# the StyleGAN2 iteration), the default bench line, the other workloads' lines, the StyleGAN2 lines + same-box A/B of this round's switches,
# the per-shape StyleGAN2 convolution table, the split-operand kernel's probe.  Output under gpurun_out/ (copied into profiles/ afterwards).
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh r06 > /dev/null 2>&1
bash $R/tools/probe/sg2_prof.sh r06 > /dev/null 2>&1
mkdir -p $R/gpurun_out/r6raw
cd $R
python bench.py > gpurun_out/r6raw/bench.json 2> gpurun_out/r6raw/bench.err
python bench.py --workload sngan64 --no_cpu_baseline 2>/dev/null | grep "^{" > gpurun_out/r6raw/bench64.json
python bench.py --phase 2 --no_cpu_baseline --no_sngan64_leg 2>/dev/null | grep "^{" > gpurun_out/r6raw/bench_p2.json
python bench.py --workload sngan64 --phase 2 --no_cpu_baseline 2>/dev/null | grep "^{" > gpurun_out/r6raw/bench64_p2.json
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline 2>/dev/null | grep "^{" > gpurun_out/r6raw/bench_dcgan.json
DIAGAN_QUIET=1 python bench.py --workload dcgan --steps 30 --warmup 5 --no_cpu_baseline --graph 2>/dev/null | grep "^{" > gpurun_out/r6raw/bench_dcgan_graph.json
python bench.py --workload stylegan2 --steps 20 --warmup 3 2>/dev/null | grep "^{" > gpurun_out/r6raw/bench_sg2.json
python bench.py --workload stylegan2 --phase 2 --steps 8 --warmup 3 2>/dev/null | grep "^{" > gpurun_out/r6raw/bench_sg2_p2.json
ALLOFF="DIAGAN_GEMM_X3B=0 DIAGAN_WGRAD_X3=0 DIAGAN_SG2_FUSED_PREP=0 DIAGAN_SG2_FUSED_SKIP=0 DIAGAN_SG2_OUT_MAP=0 DIAGAN_SG2_FUSED_TAILS=0 DIAGAN_SG2_FUSED_DENSE=0"
SG2_STEPS=16 tools/probe/sg2_ab.sh "$ALLOFF" "A=default" "DIAGAN_GEMM_X3B=0" "DIAGAN_WGRAD_X3=0" "DIAGAN_SG2_FUSED_TAILS=0" "DIAGAN_SG2_FUSED_DENSE=0" "DIAGAN_SG2_FUSED_PREP=0" "DIAGAN_SG2_FUSED_SKIP=0" "DIAGAN_GEMM_X3B_FORM=1" "A=default" "$ALLOFF" > gpurun_out/r6raw/sg2_ab.txt 2>&1
python tools/sg2_layer_times.py --iters 4 > gpurun_out/r6raw/sg2_layer_times.txt 2>&1
python tools/probe/gemm_x3b_time.py > gpurun_out/r6raw/x3b_probe.txt 2>&1
DIAGAN_WGRAD_X3_MIN_MAC=0 python tools/probe/wgrad_x3_time.py > gpurun_out/r6raw/wgrad_x3_probe.txt 2>&1
python tools/probe/dense_time.py > gpurun_out/r6raw/dense_probe.txt 2>&1
for f in bench bench64 bench_p2 bench64_p2 bench_dcgan bench_dcgan_graph bench_sg2 bench_sg2_p2; do python -c "import json,sys; d=json.loads(open('gpurun_out/r6raw/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
cat gpurun_out/r6raw/sg2_ab.txt
