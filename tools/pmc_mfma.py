#!/usr/bin/env python
"""MFMA utilisation per kernel from a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE) -> markdown.

usage: tools/pmc_mfma.py <title> <out.md> <label>=<counter_collection.csv> [<label>=<csv> ...]
Collected with (GPU box):  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv
                           -- python3 bench.py --steps 3 --warmup 2 --no_cpu_baseline --no_sngan64_leg --no_kernel_timer
Method (round 3).  GRBM_GUI_ACTIVE / 8 / wall time is the shader clock only for dispatches of >= 0.3 ms
(MI355X_MICROARCH.md, "DVFS give-back": the counter keeps running outside a short dispatch, so the quotient reads high --
round 2's tables showed 2.5-3.2 GHz for the 20-60 us kernels, above the 2.4 GHz maximum, and under-stated their MFMA-busy
by the same factor).  Here:
  * clock f of the run = sum(GRBM_GUI_ACTIVE / 8) / sum(wall time) over the dispatches of >= 0.3 ms only;
  * "MFMA busy" of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (wall time x f x 1024 SIMDs) -- wall time from the kernel trace; for
    kernels whose own dispatches are >= 0.3 ms, f is their own clock.  For short kernels the true clock lies between f
    (measured under the heaviest load) and 2.4 GHz, so the figure is an UPPER bound that can overstate by at most 2.4 / f;
  * "busy while resident" = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4 SIMDs): the share of the cycles during which a CU
    had waves at all -- insensitive to the dispatch window, but it does not see CUs left empty by a small launch (it reads
    HIGHER than "MFMA busy" for launches that do not fill the chip).  SQ_BUSY_CU_CYCLES' unit is calibrated in the table
    header against the long dispatches (cycles per CU per shader cycle: 1.0 expected)."""
import collections
import csv
import re
import sys


def short(name):
    m = re.match(r"(?:void )?diagan::([A-Za-z0-9_]+)(<[^>]*>)?", name)
    if not m:
        return None
    k = m.group(1) + (m.group(2) or "").replace(" ", "")
    # (conv_wino4_kernel's fourth template argument LEFT is 0 in every production launch: bench.py's timer name has three)
    return re.sub(r"^(conv_wino4_kernel<\d+,\d+,(?:false|true)),0>$", r"\1>", k)


def table(path):
    per = {}                   # dispatch id -> dict
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        if not k:
            continue
        d = per.setdefault(r['Dispatch_Id'], dict(k=k, t=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-9, c=collections.defaultdict(float)))
        d['c'][r['Counter_Name']] += float(r['Counter_Value'])
    long_cyc = sum(d['c'].get('GRBM_GUI_ACTIVE', 0) / 8.0 for d in per.values() if d['t'] >= 3e-4)
    long_t = sum(d['t'] for d in per.values() if d['t'] >= 3e-4)
    f = long_cyc / long_t if long_t > 0 else 2.2e9
    # unit of SQ_BUSY_CU_CYCLES: on long full-chip dispatches  counter / (256 CUs x shader cycles)
    cal_num = sum(d['c'].get('SQ_BUSY_CU_CYCLES', 0) for d in per.values() if d['t'] >= 3e-4)
    cal = cal_num / (256.0 * long_cyc) if long_cyc > 0 else float('nan')
    agg = {}
    for d in per.values():
        if not d['k'].startswith('conv'):
            continue
        a = agg.setdefault(d['k'], dict(n=0, t=0.0, mfma=0.0, cu=0.0, grbm=0.0, tl=0.0, gl=0.0))
        a['n'] += 1
        a['t'] += d['t']
        a['mfma'] += d['c'].get('SQ_VALU_MFMA_BUSY_CYCLES', 0)
        a['cu'] += d['c'].get('SQ_BUSY_CU_CYCLES', 0)
        if d['t'] >= 3e-4:
            a['tl'] += d['t']
            a['gl'] += d['c'].get('GRBM_GUI_ACTIVE', 0) / 8.0
    rows = []
    for k, a in agg.items():
        own = a['tl'] >= 0.5 * a['t'] and a['tl'] > 0          # most of the kernel's time is in long dispatches: its own clock
        fk = a['gl'] / a['tl'] if own else f
        busy = a['mfma'] / (a['t'] * fk * 1024.0)
        resident = a['mfma'] / (a['cu'] / cal * 4.0) if a['cu'] > 0 and cal == cal and cal > 0 else float('nan')
        rows.append((a['t'], k, a['n'], busy, resident, fk / 1e9, own, a['t'] / a['n'] * 1e6))
    return sorted(rows, reverse=True), f, cal


def main():
    title, out = sys.argv[1:3]
    with open(out, 'w') as fo:
        fo.write(f"# {title}\n\n" + __doc__.split('usage:')[1].split('\n', 1)[1] + "\n")
        for spec in sys.argv[3:]:
            label, path = spec.rsplit('=', 1)
            rows, f, cal = table(path)
            fo.write(f"\n## {label}\n\nclock of the run (dispatches >= 0.3 ms): {f / 1e9:.2f} GHz; SQ_BUSY_CU_CYCLES per CU per shader cycle on "
                     f"those dispatches: {cal:.2f}\n\n| kernel | launches | avg us | total ms | MFMA busy | busy while resident | clock used (GHz) |\n"
                     "|---|---|---|---|---|---|---|\n")
            for t, k, n, u, res, clk, own, avg in rows[:14]:
                fo.write(f"| `{k}` | {n} | {avg:.0f} | {t * 1e3:.2f} | {u:.3f}{'' if own else ' (upper bound)'} | {res:.3f} | "
                         f"{clk:.2f}{' (own)' if own else ' (run)'} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
