#!/usr/bin/env python
"""profiles/r02_sngan{32,64}_summary.md (+ kernel_stats CSVs, PMC tables) from a tools/profile_round.sh output directory and
the JSON line of an un-profiled default `python bench.py` run.
usage: tools/round_docs.py gpurun_out/prof_<tag> <bench.json>"""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof, bench_json = sys.argv[1], sys.argv[2]


def line(path):
    for l in open(path):
        if l.startswith('{'):
            return json.loads(l)


def table(csv, steps):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools/prof_summary.py"), csv, str(steps), "/tmp/_t.md", "t"],
                         capture_output=True, text=True, check=True)
    return open("/tmp/_t.md").read().split('\n', 2)[2]


d = line(bench_json)
b32, b64 = line(f"{prof}/kt32.log"), line(f"{prof}/kt64.log")
c32, c64 = glob.glob(f"{prof}/kt32/*/*kernel_stats.csv")[0], glob.glob(f"{prof}/kt64/*/*kernel_stats.csv")[0]
shutil.copy(c32, os.path.join(ROOT, "profiles/r02_sngan32_kernel_stats.csv"))
shutil.copy(c64, os.path.join(ROOT, "profiles/r02_sngan64_kernel_stats.csv"))
r, s = d['roofline'], d['sngan64_conv_blocks']
open(os.path.join(ROOT, "profiles/r02_sngan32_summary.md"), 'w').write(f"""# r02: SNGAN-32 bs=64 phase-1 step (BASELINE configs[1]), final round-2 build

Command (GPU box, `tools/profile_round.sh`): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5
--warmup 2 --no_cpu_baseline --no_x6_leg --no_sngan64_leg` (2 warm-up + 5 timed + 2 un-timed table steps = 9 global steps
profiled; {b32['value']} images/s under the profiler).

Un-profiled default run of the same build (`python bench.py`): **{d['value']} images/s, {d['ms_per_step']} ms/step**
(round 1: 2421 / 26.4).  Roofline kernel `{r['kernel']}` (Winograd F(2x2,3x3) forward / data-gradient): HIP-event average
{r['avg_launch_us']} us/launch over the {r['launches']} launches of the timed region (the rocprof average below runs over all 9
steps incl. the smaller launches of the table steps), {r['algorithmic_gflop_per_launch']} algorithmic GFLOP/launch (direct
convolution, 2 M Co 9 Ci) = {r['algorithmic_tflops']} TFLOP/s; the kernel EXECUTES 16/36 of that =
{r['executed_gflop_per_launch']} GFLOP/launch -> **{r['achieved']} TFLOP/s = {r['frac']} of the 157.3 TFLOP/s fp32 MFMA peak**
(PMC pass: `r02_mfma_util.md`).  What bounds it and how it got here from 0.59: `r02_wino_ablation.md`.
Whole-step algorithmic rate: 2.871 TFLOP / {d['ms_per_step']} ms = {2871 / d['ms_per_step']:.0f} TFLOP/s
({2871 / d['ms_per_step'] / 157.3:.2f} of the fp32 MFMA peak in direct-convolution FLOP; round 1: 107).

SNGAN-64 leg of the same default run (`sngan64_conv_blocks`): {s['images_per_s']} images/s, {s['ms_per_step']} ms/step; residual-block
convolutions {s['tflops']} TFLOP/s algorithmic = **{s['frac']} of peak** (north_star bar: 0.60), {s['mfma_executed_tflops']}
executed ({s['mfma_executed_frac']}).

""" + table(c32, 9))
open(os.path.join(ROOT, "profiles/r02_sngan64_summary.md"), 'w').write(f"""# r02: SNGAN-64 (CelebA configuration) bs=64 phase-1 step, final round-2 build (VERDICT r1 item 1)

Command (GPU box, `tools/profile_round.sh`): `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload
sngan64 --steps 5 --warmup 2 --no_cpu_baseline --no_x6_leg` (9 global steps profiled; {b64['value']} images/s under the
profiler, {s['images_per_s']} in the un-profiled default run; round 1: 1319).

north_star's kernel target -- >= 60 % of the MFMA roofline on the SNGAN 64x64 conv blocks at bs = 64 -- is emitted by the
DEFAULT `python bench.py` run as `sngan64_conv_blocks` (definition in the JSON line and DESIGN 6):
{s['conv_block_gflop_per_step']} algorithmic GFLOP of residual-block convolutions per step (a4 + a5, forward + both gradients;
l1, c6, head excluded) over {s['conv_block_kernel_ms_per_step']} ms of their kernels = **{s['tflops']} TFLOP/s = {s['frac']} of
157.3**; counting the Winograd launches at the 16/36 of the multiply-accumulates they execute: {s['mfma_executed_tflops']}
TFLOP/s = {s['mfma_executed_frac']}.  The rocprof table below gives the same kernels' ms/step; MFMA-busy PMC table:
`r02_mfma_util.md`.

""" + table(c64, 9))
print(open(os.path.join(ROOT, "profiles/r02_sngan32_summary.md")).read()[:1500])
