#!/usr/bin/env python
"""Soak (GPU box): N global steps of the SNGAN-32 phase-1 (or --phase 2) step of bench.py on fresh synthetic batches, printing
every 100 steps the device memory in use / reserved, the host RSS and whether every parameter is finite -- what the two-to-six-step
end-to-end tests cannot see (a buffer list that grows, a cache keyed by something that changes every step).
usage: tools/soak.py [--steps 1000] [--dataset cifar10|celeba] [--phase 1|2]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import psutil
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--dataset", default="cifar10")
    ap.add_argument("--phase", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    res = 32 if a.dataset == "cifar10" else 64
    netG, netD, netD_drs, optG, optD, optD_drs = bench.build_models(a.dataset, "ns", a.phase, dev)
    g = torch.Generator(device=dev).manual_seed(3)
    batches = [torch.randn(64, 3, res, res, device=dev, generator=g).clamp_(-1, 1) for _ in range(16)]
    step = bench.make_global_step(netG, netD, netD_drs, optG, optD, optD_drs, batches, 5, a.steps, dev)
    proc = psutil.Process()
    base = None
    for i in range(a.steps + 1):
        if i % 100 == 0:
            torch.cuda.synchronize()
            fin = all(torch.isfinite(p).all().item() for n in (netG, netD) for p in n.parameters())
            al, rs, rss = torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20, proc.memory_info().rss / 2**20
            if i == 200:
                base = (al, rss)
            print(f"step {i:5d}: device allocated {al:8.1f} MiB, reserved {rs:8.1f} MiB, host RSS {rss:8.1f} MiB, parameters finite: {fin}", flush=True)
            assert fin, "non-finite parameter"
        if i < a.steps:
            step()
    if base:
        print(f"growth since step 200: device {al - base[0]:+.1f} MiB, host {rss - base[1]:+.1f} MiB")


if __name__ == "__main__":
    main()
