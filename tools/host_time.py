#!/usr/bin/env python
"""How far ahead of the GPU does the host run?  (GPU box)  Times the launch loop of N global steps
without synchronising, then the drain; and a cProfile of the host side of the steps."""
import os, sys, time, cProfile, pstats, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'sngan32'
dataset, res, _ = bench.WORKLOADS[wl]
dev = torch.device('cuda', 0)
nets = bench.build_models(dataset, 'ns', 1, dev)
batches = [(torch.rand(64, 3, res, res) * 2 - 1).to(dev) for _ in range(10)]
step = bench.make_global_step(*nets, batches, 5, 50000, dev)
for _ in range(5): step()
torch.cuda.synchronize()
N = 20
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host launch loop {1e3*(t1-t0)/N:.2f} ms/step, drain {1e3*(t2-t1):.2f} ms total, wall {1e3*(t2-t0)/N:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(5): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])
