#!/usr/bin/env python
"""Which ATen operators (fills, copies, random draws) does an SNGAN global step still launch, and from which source lines?
torch.profiler with stacks over three global steps of bench.py's step (GPU box).   python tools/sngan_aten_ops.py [sngan32|sngan64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

wl = sys.argv[1] if len(sys.argv) > 1 else 'sngan32'
dataset, res, desc = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
nets = bench.build_models(dataset, 'ns', 1, dev)
gen = torch.Generator().manual_seed(1234)
batches = [(torch.rand(64, 3, res, res, generator=gen) * 2 - 1).to(dev) for _ in range(10)]
step = bench.make_global_step(*nets, batches, 5, num_steps=50000, device=dev)
for _ in range(4):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = {}
for e in prof.events():
    if not e.name.startswith("aten::"):
        continue
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt <= 0:
        continue
    site = next((s for s in (e.stack or []) if "self-diagnosing-gan_amd" in s or "bench.py" in s), "?")
    site = site.replace(ROOT + "/", "")
    key = (e.name, str(e.input_shapes)[:70], site[:110])
    r = rows.setdefault(key, [0.0, 0])
    r[0] += dt / N / 1e3
    r[1] += 1.0 / N
tot = sum(v[0] for v in rows.values())
print(f"{desc}: aten:: operators with device time: {tot:.3f} ms per global step, {sum(v[1] for v in rows.values()):.1f} launches")
for (name, shp, site), (ms, cnt) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
    print(f"{ms:7.4f} ms {cnt:5.1f} x {name:22s} {shp:70s} {site}")
