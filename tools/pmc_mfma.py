#!/usr/bin/env python
"""MFMA utilisation per kernel from a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) -> markdown.

usage: tools/pmc_mfma.py <title> <out.md> <label>=<counter_collection.csv> [<label>=<csv> ...]
Collected with (GPU box):  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv
                           -- python3 bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_x6_leg --no_kernel_timer
utilisation = MFMA busy cycles / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (summed over the 8 XCDs);
effective clock = kernel cycles / kernel time."""
import collections
import csv
import re
import sys


def short(name):
    m = re.match(r"(?:void )?diagan::([A-Za-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "").replace(" ", "")) if m else None


def table(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt, tim, seen = collections.Counter(), collections.defaultdict(float), set()
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        if not k or not k.startswith('conv'):
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            cnt[k] += 1
            tim[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9
    rows = []
    for k, v in acc.items():
        cycles = v.get('GRBM_GUI_ACTIVE', 0) / 8.0
        if cycles > 0:
            rows.append((tim[k], k, cnt[k], v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cycles * 1024), cycles / tim[k] / 1e9))
    return sorted(rows, reverse=True)


def main():
    title, out = sys.argv[1:3]
    with open(out, 'w') as f:
        f.write(f"# {title}\n\n" + __doc__.split('usage:')[1].split('\n', 1)[1] + "\n")
        for spec in sys.argv[3:]:
            label, path = spec.rsplit('=', 1)
            f.write(f"\n## {label}\n\n| kernel | launches | total ms | MFMA busy | clock GHz |\n|---|---|---|---|---|\n")
            for t, k, n, u, clk in table(path)[:10]:
                f.write(f"| `{k}` | {n} | {t * 1e3:.2f} | {u:.3f} | {clk:.2f} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
