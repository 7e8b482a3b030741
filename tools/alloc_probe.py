"""Does the steady-state step hit the device allocator? (GPU box)  Prints per-step wall time and the change in
torch's caching-allocator segment counters for the G and D updates of a workload."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'sngan64'
dataset, res, _ = bench.WORKLOADS[wl]
dev = torch.device('cuda', 0)
nets = bench.build_models(dataset, 'ns', 1, dev)
batches = [(torch.rand(64, 3, res, res) * 2 - 1).to(dev) for _ in range(10)]
step = bench.make_global_step(*nets, batches, 5, 50000, dev)
for _ in range(3): step()
torch.cuda.synchronize()
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "segment.all.allocated", "reserved_bytes.all.current")
for i in range(6):
    s0 = torch.cuda.memory_stats()
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); t1 = time.perf_counter()
    s1 = torch.cuda.memory_stats()
    print(f"step {i}: {1e3*(t1-t0):7.2f} ms", {k: s1.get(k, 0) - s0.get(k, 0) for k in keys[:4]}, "reserved GB", round(s1[keys[4]] / 2**30, 2))
