#!/usr/bin/env python
"""rocprofv3 --kernel-trace --stats CSV -> markdown summary under profiles/.
usage: tools/prof_summary.py <kernel_stats.csv> <steps profiled> <out.md> [title]"""
import csv
import sys

src, steps, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
title = sys.argv[4] if len(sys.argv) > 4 else out
rows = list(csv.DictReader(open(src)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
gemm = sum(float(r['TotalDurationNs']) for r in rows if 'conv_gemm' in r['Name'] or 'conv_wgrad' in r['Name']
           or 'conv3x3_co4' in r['Name'] or 'conv_wino' in r['Name'] or 'wino_weight' in r['Name'])
with open(out, 'w') as f:
    f.write(f"# {title}\n\n")
    f.write(f"Total kernel time {tot / steps / 1e6:.2f} ms per global step ({steps:g} steps profiled); "
            f"conv GEMM kernels {gemm / steps / 1e6:.2f} ms ({gemm / tot:.1%}).\n\n")
    f.write("| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|\n")
    for r in rows:
        if float(r['Percentage']) < 0.05:
            continue
        f.write(f"| `{r['Name'][:110]}` | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / steps / 1e6:.3f} | "
                f"{float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |\n")
print(open(out).read()[:6000])
