#!/usr/bin/env python
"""How far ahead of the GPU does the host run?  (GPU box)  Times the launch loop of N global steps
without synchronising, then the drain; optionally a cProfile of the host side of the steps.

    python tools/host_time.py [sngan32|sngan64|dcgan] [--phase 2] [--cores C] [--profile]

--cores C pins the process to C cores BEFORE torch is imported (what a rank gets when W ranks share a
cgroup quota: 16 threads / 8 ranks = 2) and sets the torch / OMP thread counts accordingly."""
import argparse, os, sys, time
ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="sngan32")
ap.add_argument("--phase", type=int, default=1)
ap.add_argument("--cores", type=int, default=0)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
if args.cores > 0:
    avail = sorted(os.sched_getaffinity(0))
    os.sched_setaffinity(0, set(avail[:args.cores]))
    os.environ["OMP_NUM_THREADS"] = str(args.cores)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import bench

wl = args.workload
dataset, res, _ = bench.WORKLOADS[wl]
dev = torch.device('cuda', 0)
nets = bench.build_models(dataset, 'ns', args.phase, dev)
batches = [(torch.rand(64, 3, res, res) * 2 - 1).to(dev) for _ in range(10)]
step = bench.make_global_step(*nets, batches, 5, 50000, dev)
for _ in range(5): step()
torch.cuda.synchronize()
N = args.steps
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
host, wall = 1e3 * (t1 - t0) / N, 1e3 * (t2 - t0) / N
print(f"HOST_TIME workload={wl} phase={args.phase} cores={len(os.sched_getaffinity(0))} "
      f"host_launch_loop_ms_per_step={host:.2f} drain_ms_total={1e3*(t2-t1):.2f} wall_ms_per_step={wall:.2f} "
      f"host_over_wall={host/wall:.3f}")
if args.profile:
    import cProfile, pstats, io
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5): step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
    print(s.getvalue()[:6000])
