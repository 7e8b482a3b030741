#!/usr/bin/env python
"""Which ATen operators make up the element-wise glue of a StyleGAN2 iteration (256^2, batch 32)?  torch.profiler over four
iterations, operators with device time, grouped by name and input shapes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
step = bench.make_stylegan2_step(256, 32, 1, dev)
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt / N / 1e3, e.count / N, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"device time of all operators: {tot:.1f} ms per iteration")
aten = [r for r in rows if r[2].startswith("aten::")]
print(f"aten:: operators (self device time): {sum(r[0] for r in aten):.1f} ms per iteration")
for ms, cnt, key, shp in aten[:45]:
    print(f"{ms:8.3f} ms  {cnt:7.1f} calls  {key:28s} {shp}")
if os.environ.get("SG2_BY_COUNT") == "1":        # the launch-count view: which small operators are called most
    print("aten:: operators by calls per iteration:")
    for ms, cnt, key, shp in sorted(aten, key=lambda r: -r[1])[:70]:
        print(f"{cnt:7.1f} calls  {ms:8.3f} ms  {key:28s} {shp}")
own = [r for r in rows if not r[2].startswith("aten::") and not r[2].startswith("void ") and "diagan::" not in r[2]]
print("autograd Functions of this engine (self device time):")
for ms, cnt, key, shp in own[:12]:
    print(f"{ms:8.3f} ms  {cnt:7.1f} calls  {key:28s} {shp}")
