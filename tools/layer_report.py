#!/usr/bin/env python
"""Per-shape GEMM report of one global step (GPU box): which layers run below the roofline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import bench
from diagan.ops import conv as C

wl = sys.argv[1] if len(sys.argv) > 1 else 'sngan32'
dataset, res, _ = bench.WORKLOADS[wl]
dev = torch.device('cuda', 0)
nets = bench.build_models(dataset, 'ns', 1, dev)
batches = [(torch.rand(64, 3, res, res) * 2 - 1).to(dev) for _ in range(10)]
step = bench.make_global_step(*nets, batches, 5, 50000, dev)
for _ in range(2): step()
torch.cuda.synchronize()
C.TIMER = C.KernelTimer()
for _ in range(3): step()
torch.cuda.synchronize()
rows = sorted(C.TIMER.by_shape().items(), key=lambda kv: -kv[1]['seconds'])
tot = sum(v['seconds'] for _, v in rows)
print(f"total GEMM time {tot/3*1e3:.2f} ms/step")
for k, v in rows[:100]:
    print(f"{k[0]:32s} M={k[1]:7d} N={k[2]:5d} K={k[3]:5d} {k[4]:16s} n/step {v['launches']/3:5.1f} "
          f"avg {v['seconds']/v['launches']*1e6:7.1f} us  {v['flop']/v['seconds']/1e12:6.1f} TF  "
          f"share {v['seconds']/tot:5.1%}")
