#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes) -> profiles/pmc_traffic.json (read by bench.py) + a markdown table.

usage: tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <workload> <out.md> [grbm csv]
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts 128-byte requests at 64 B on gfx950
(doubled for 16-B/lane coalesced reads); both counters are in KiB."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.match(r"(?:void )?diagan::([A-Za-z0-9_]+)(<[^>]*>)?", name)
    if not m:
        return None
    return m.group(1) + (m.group(2) or "").replace(" ", "")


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        a = acc[k]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    return acc


def main():
    fetch, write, workload, out = sys.argv[1:5]
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    clock = {}
    if len(sys.argv) > 5:
        for k, (v, n, t) in per_kernel(sys.argv[5], "GRBM_GUI_ACTIVE").items():
            clock[k] = v / 8.0 / t / 1e9 if t > 0 else None     # sum over 8 XCDs -> GHz
    rows = []
    for k in f:
        if k not in w or not (k.startswith("conv_") or k.startswith("conv3x3")):
            continue
        fk, wk = f[k][0] / f[k][1], w[k][0] / w[k][1]
        rows.append((k, f[k][1], fk, wk, (2 * fk + wk) * 1024, clock.get(k)))
    rows.sort(key=lambda r: -r[4] * r[1])
    jpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    data = json.load(open(jpath)) if os.path.exists(jpath) else {}
    data[workload] = {r[0]: int(r[4]) for r in rows}
    data["_note"] = ("HBM bytes per launch (average over the launches of the profiled bench.py run) = (2*FETCH_SIZE + "
                     "WRITE_SIZE)*1024: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B/lane coalesced "
                     "reads on gfx950; FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes.")
    json.dump(data, open(jpath, "w"), indent=1)
    with open(out, "w") as fo:
        fo.write(f"# PMC traffic, {workload} (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE"
                 f"{' / GRBM_GUI_ACTIVE' if clock else ''}, separate passes)\n\n")
        fo.write("| kernel | launches | FETCH_SIZE avg (KiB, raw) | WRITE_SIZE avg (KiB) | HBM MB/launch = (2F+W) | "
                 "effective clock GHz |\n|---|---|---|---|---|---|\n")
        for k, n, fk, wk, b, c in rows:
            fo.write(f"| `{k}` | {n} | {fk:.0f} | {wk:.0f} | {b / 1e6:.1f} | {'' if c is None else f'{c:.2f}'} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
