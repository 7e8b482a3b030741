# rocprofv3 kernel stats of the StyleGAN2 256^2 iteration (GPU box) -> gpurun_out/prof_sg2/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_sg2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload stylegan2 --steps 6 --warmup 2 --no_cpu_baseline > $O/kt.log 2>&1
find $O/kt -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
tail -1 $O/kt.log | cut -c1-300
