#!/usr/bin/env python
"""Where the launch thread's time goes (GPU box): cProfile over N global steps of bench.py's step, phase 1 or 2.
   python tools/probe/host_profile.py [sngan32|sngan64] [1|2]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else 'sngan32'
phase = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dataset, res, _ = bench.WORKLOADS[wl]
dev = torch.device('cuda', 0)
nets = bench.build_models(dataset, 'ns', phase, dev)
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
print(f"{wl} phase {phase}: launch loop {(t1 - t0) / N * 1e3:.2f} ms/step, with the final sync {(t2 - t0) / N * 1e3:.2f}")
if os.environ.get("HOST_TIMER") == "1":      # as bench.py's timed region: the kernel timer brackets the dominant kernel only
    from diagan.ops import conv as C
    C.TIMER = C.KernelTimer(only={"conv_wino4_kernel<2,0,false>"})
pr = cProfile.Profile()
pr.enable()
for _ in range(N): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats('tottime')
ps.print_stats(45)
out = s.getvalue().replace(ROOT + "/", "")
print(out[:9000])
