mkdir -p gpurun_out/r3e
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3e/gpu_tests.txt
python bench.py --steps 20 --warmup 5 --no_x6_leg > gpurun_out/r3e/bench.json 2> gpurun_out/r3e/bench.err
python bench.py --workload sngan32 --phase 2 --steps 10 --warmup 3 --no_x6_leg --no_cpu_baseline --no_sngan64_leg > gpurun_out/r3e/bench_p2.json 2> gpurun_out/r3e/bench_p2.err
cat gpurun_out/r3e/gpu_tests.txt
python - <<'PY'
import json
for f in ("bench", "bench_p2"):
    try:
        d = json.loads(open(f"gpurun_out/r3e/{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("sngan64_conv_blocks", {}).get("images_per_s"), d.get("sngan64_conv_blocks", {}).get("mfma_executed_frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
