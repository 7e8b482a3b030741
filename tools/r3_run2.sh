mkdir -p gpurun_out/r3d
python -m pytest tests/test_wino4_gpu.py tests/test_wino_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r3d/wino_tests.txt
python -m pytest tests/test_sngan_gpu.py tests/test_conv_gpu.py tests/test_e2e_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r3d/sngan_tests.txt
python bench.py --steps 20 --warmup 5 --no_x6_leg > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
DIAGAN_WINO4=0 python bench.py --steps 20 --warmup 5 --no_x6_leg --no_cpu_baseline > gpurun_out/r3d/bench_now4.json 2> gpurun_out/r3d/bench_now4.err
cat gpurun_out/r3d/wino_tests.txt gpurun_out/r3d/sngan_tests.txt
python - <<'PY'
import json
for f in ("bench", "bench_now4"):
    try:
        d = json.loads(open(f"gpurun_out/r3d/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("sngan64_conv_blocks", {}).get("images_per_s"), d.get("sngan64_conv_blocks", {}).get("mfma_executed_frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
