#!/usr/bin/env python
"""the CPU restatement (oracle/nets.py, oracle/scorer.py) timed on the GPU box's host cores.
Bounded samples (a few updates per row); one global step = n_dis * t_D [+ n_dis * t_D for D_drs] + t_G."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import nets as O
from oracle import scorer as osc
import bench

avail = len(os.sched_getaffinity(0))
try:
    quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
    if quota != 'max':
        avail = max(1, min(avail, int(quota) // int(period)))
except (OSError, ValueError):
    pass
cores = min(avail, 64)
torch.set_num_threads(cores)
print(f"CPU: {bench.cpu_model_name()}, {cores} threads of {avail} usable")


def row(tag, dataset, res, n_dis, drs, reps):
    oG, oD, ooptG, ooptD = O.make_pair(dataset, 'ns', seed=1)
    x = torch.rand(64, 3, res, res, generator=torch.Generator().manual_seed(1)) * 2 - 1
    oD.train_step((x, None), oG, ooptD)
    td, tg = [], []
    for _ in range(reps):
        t0 = time.time(); oD.train_step((x, None), oG, ooptD); td.append(time.time() - t0)
        t0 = time.time(); oG.train_step((x, None), oD, ooptG); tg.append(time.time() - t0)
    t_d, t_g = min(td), min(tg)
    step = n_dis * t_d * (2 if drs else 1) + t_g
    print(f"{tag}: t_D {t_d:.3f} s, t_G {t_g:.3f} s -> {step:.2f} s per global step = {64 / step:.2f} images/s", flush=True)


row("C1 mnist_dcgan phase 1 (n_dis=1)", 'color_mnist', 32, 1, False, 5)
row("C2 SNGAN-32 phase 1", 'cifar10', 32, 5, False, 3)
row("C3 SNGAN-32 phase 2 (+D_drs)", 'cifar10', 32, 5, True, 2)
row("C4 SNGAN-64 phase 1", 'celeba', 64, 5, False, 1)
row("C4 SNGAN-64 phase 2 (+D_drs)", 'celeba', 64, 5, True, 1)
for N in (50000, 162770):
    rng = np.random.default_rng(0)
    logits = {100 * t: rng.normal(size=N).astype(np.float32).astype(np.float64) for t in range(50)}
    t0 = time.time(); osc.calculate_scores_numpy(logits, 0, 10 ** 9); t_np = time.time() - t0
    t0 = time.time(); osc.calculate_scores_c(logits, 0, 10 ** 9); t_c = time.time() - t0
    print(f"C5 calculate_scores T=50 N={N}: NumPy restatement {t_np * 1e3:.0f} ms, C oracle {t_c * 1e3:.0f} ms (1 thread)", flush=True)
