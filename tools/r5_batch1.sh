#!/bin/bash
# round 5, GPU-box batch 1: new tests, bench legs, f64 ratios, 20-step trajectory
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests/test_sngan_gpu.py -x -q -s -k "float64 or deep_copy" 2>&1 | grep -E "f64-parity|passed|failed|Error|error" | tee gpurun_out/r5/f64_ratios.txt
timeout 600 python -m pytest tests/test_e2e_gpu.py tests/test_wino4_gpu.py -x -q -k "logit or x3" 2>&1 | tail -5 | tee gpurun_out/r5/new_tests.txt
timeout 300 python bench.py > gpurun_out/r5/bench.json 2> gpurun_out/r5/bench.err; tail -c 1500 gpurun_out/r5/bench.json | head -c 1500; echo
timeout 1500 python tools/sngan_trajectory.py 20 cifar10 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5/trajectory_cifar10.txt | tail -8
