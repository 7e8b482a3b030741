#!/bin/bash
mkdir -p gpurun_out/r5
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r5/gpu_suite_final.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 200 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_ratio'], d['phase2']['images_per_s'], d['logit_pass']['images_per_s'], d['sngan64_conv_blocks']['images_per_s'], d['sngan64_conv_blocks']['frac'])"
