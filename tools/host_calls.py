#!/usr/bin/env python
"""Where does the host time of a global step go?  (GPU box)  Wraps diagan._native.call with a wall-clock accumulator
per entry point and times torch.empty / torch.zeros; prints per-name totals over N un-synchronised steps."""
import argparse, os, sys, time, collections
ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="sngan32")
ap.add_argument("--phase", type=int, default=1)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import bench
from diagan import _native as nat

dataset, res, _ = bench.WORKLOADS[args.workload]
dev = torch.device('cuda', 0)
nets = bench.build_models(dataset, 'ns', args.phase, dev)
batches = [(torch.rand(64, 3, res, res) * 2 - 1).to(dev) for _ in range(10)]
step = bench.make_global_step(*nets, batches, 5, 50000, dev)
for _ in range(5): step()
torch.cuda.synchronize()
acc = collections.defaultdict(lambda: [0, 0.0])
pc = time.perf_counter
orig_fn = nat.fn
def timed_call(name, *a):
    f = orig_fn(name)
    t0 = pc()
    rc = f(*a)
    dt = pc() - t0
    e = acc[name]; e[0] += 1; e[1] += dt
    if rc != 0:
        raise RuntimeError(name)
nat.call = timed_call
import diagan.ops.conv as C, diagan.ops.eltwise as E
for mod in (C, E):
    if hasattr(mod, 'nat'):
        mod.nat.call = timed_call
for name in ("empty", "zeros", "empty_like", "zeros_like", "cat", "randn"):
    def mk(name, f):
        def w(*a, **k):
            t0 = pc(); r = f(*a, **k); dt = pc() - t0
            e = acc["torch." + name]; e[0] += 1; e[1] += dt
            return r
        return w
    setattr(torch, name, mk(name, getattr(torch, name)))
N = args.steps
t0 = pc()
for _ in range(N): step()
t1 = pc()
torch.cuda.synchronize()
tot = sum(v[1] for v in acc.values())
print(f"HOST_CALLS workload={args.workload} phase={args.phase}: loop {1e3*(t1-t0)/N:.2f} ms/step, inside native+torch calls {1e3*tot/N:.2f} ms/step, "
      f"{sum(v[0] for v in acc.values())/N:.0f} calls/step")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {k:40s} {v[0]/N:7.1f} calls/step  {1e6*v[1]/v[0]:8.2f} us/call  {1e3*v[1]/N:7.3f} ms/step")
