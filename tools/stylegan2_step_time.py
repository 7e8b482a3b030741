#!/usr/bin/env python
"""Time the StyleGAN2 training iteration (BASELINE config 5: 256x256, batch 32 per GPU) on one MI355X.

    python tools/stylegan2_step_time.py [--size 256] [--batch 32] [--iters 16] [--phase2]

One "iteration" = the reference loop body (train_ffhq.py:199-300): D step, G step, EMA; every 16th iteration also the
R1 step, every 4th the path-length step.  Reports the mean over a whole number of 16-iteration cycles, plus the
plain / regularised iteration times, and (with --kernels) the per-kernel HIP-event table of the GEMM launches."""
import argparse
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))

import torch  # noqa: E402


class Synthetic(torch.utils.data.Dataset):
    def __init__(self, n, size):
        self.x = torch.randn(n, 3, size, size).clamp_(-1, 1)

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], i


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--phase2", action="store_true")
    ap.add_argument("--kernels", action="store_true")
    a = ap.parse_args()
    from diagan.models.stylegan2 import StyleGANDiscriminator, StyleGANGenerator
    from diagan.ops import conv as K
    from diagan.trainer import distributed as dist
    from diagan.trainer import stylegan2 as TR
    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N tools/stylegan2_step_time.py`
    rank, local_rank, world = dist.init_from_env()
    dev = torch.device("cuda", local_rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    G, D = StyleGANGenerator(size=a.size).to(dev), StyleGANDiscriminator(size=a.size).to(dev)
    g_ema = StyleGANGenerator(size=a.size).to(dev).eval()
    TR.accumulate(g_ema, G, 0)
    g_optim, d_optim = TR.make_optimizers(G, D)
    args = types.SimpleNamespace(iter=10 ** 9, start_iter=0, batch=a.batch, latent=512, mixing=0.9, r1=10.0,
                                 d_reg_every=16, g_reg_every=4, path_regularize=2.0, path_batch_shrink=2,
                                 logit_save_steps=10 ** 9, save_logit_after=10 ** 9, stop_save_logit_after=0,
                                 n_sample=16, augment=False)
    ds = Synthetic(a.batch * 4, a.size)
    mk = lambda: torch.utils.data.DataLoader(ds, batch_size=a.batch, shuffle=True, drop_last=True)
    extra = {}
    if a.phase2:
        D2 = StyleGANDiscriminator(size=a.size).to(dev)
        extra = dict(drs_loader=mk(), drs_discriminator=D2, drs_d_optim=TR.make_optimizers(G, D2)[1])
    tr = TR.StyleGAN2Trainer(args, mk(), G, D, g_optim, d_optim, g_ema, dev, "/tmp/sg2_time", **extra)
    zero = torch.tensor(0.0, device=dev)
    tr.r1_loss, tr.path_loss, tr.path_lengths = zero, zero, zero
    if rank == 0:
        print(f"params: G {sum(p.numel() for p in G.parameters()) / 1e6:.1f} M, D "
              f"{sum(p.numel() for p in D.parameters()) / 1e6:.1f} M; {world} rank(s)", flush=True)
    for i in range(1, a.warmup + 1):          # iteration numbers that do not trigger the regularisers... except 4
        tr.train_step(i)
    dist.synchronize()
    torch.cuda.synchronize()
    per = {}
    if a.kernels:
        K.TIMER = K.KernelTimer()
    t_all = time.perf_counter()
    for i in range(a.iters):
        t0 = time.perf_counter()
        losses = tr.train_step(i)
        torch.cuda.synchronize()
        kind = ("r1+" if i % 16 == 0 else "") + ("path" if i % 4 == 0 else "") or "plain"
        per.setdefault(kind, []).append(time.perf_counter() - t0)
    dist.synchronize()
    total = time.perf_counter() - t_all
    if world > 1:
        t = torch.tensor([total], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        total = float(t)
    if rank != 0:
        return
    print({k: float(v.detach()) for k, v in losses.items()})
    for k, v in per.items():
        print(f"{k:8s}: {1e3 * sum(v) / len(v):8.1f} ms  (n={len(v)})")
    print(f"mean iteration {1e3 * total / a.iters:.1f} ms = {world * a.batch * a.iters / total:.1f} images/s "
          f"(size {a.size}, batch {a.batch} per GPU x {world} GPU(s), phase {'2' if a.phase2 else '1'}); "
          f"peak memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")
    if a.kernels:
        rows = sorted(K.TIMER.by_shape().items(), key=lambda kv: -kv[1]['seconds'])
        tot = sum(d['seconds'] for _, d in rows)
        print(f"GEMM launches: {1e3 * tot / a.iters:.1f} ms / iteration")
        for key, d in rows[:40]:
            print(f"{1e3 * d['seconds'] / a.iters:8.2f} ms  n={d['launches'] / a.iters:6.1f}  "
                  f"{d['flop'] / d['seconds'] / 1e12:6.1f} TF  {key}")


if __name__ == "__main__":
    main()
