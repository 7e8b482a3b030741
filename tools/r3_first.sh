mkdir -p gpurun_out/r3a
for c in 0 2 1; do for w in "sngan32" "sngan64" "sngan32 --phase 2"; do
python tools/host_time.py $w --cores $c 2>/dev/null | grep HOST_TIME >> gpurun_out/r3a/host_time.txt
done; done
python tools/host_time.py sngan32 --cores 2 --profile > gpurun_out/r3a/host_profile_2c.txt 2>&1
python bench.py --steps 20 --warmup 5 --no_x6_leg > gpurun_out/r3a/bench_head.json 2> gpurun_out/r3a/bench_head.err
cat gpurun_out/r3a/host_time.txt
