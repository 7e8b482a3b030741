# round-3 GPU check: DP tests, stacked-forward tests, host time, bench
mkdir -p gpurun_out/r3b
python -m pytest tests/test_dp_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r3b/dp_tests.txt
python -m pytest tests/test_sngan_gpu.py -x -q -k "stacked or prefetched" 2>&1 | tail -15 > gpurun_out/r3b/stack_tests.txt
for w in "sngan32" "sngan64" "sngan32 --phase 2"; do
python tools/host_time.py $w --cores 2 2>/dev/null | grep HOST_TIME >> gpurun_out/r3b/host_time.txt
done
python bench.py --steps 20 --warmup 5 --no_x6_leg > gpurun_out/r3b/bench.json 2> gpurun_out/r3b/bench.err
cat gpurun_out/r3b/dp_tests.txt gpurun_out/r3b/stack_tests.txt gpurun_out/r3b/host_time.txt
