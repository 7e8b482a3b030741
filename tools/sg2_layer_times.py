#!/usr/bin/env python
"""Per-shape table of the convolution launches of StyleGAN2 training iterations (256 x 256, batch 32): every call of
diagan.ops.conv.conv_fwd / conv_dgrad / conv_wgrad made by the autograd ops is timed with HIP events (synchronously, so
the sum is a little above the asynchronous iteration) and grouped by (op, geometry, shape).
    python tools/sg2_layer_times.py [--size 256] [--batch 32] [--iters 4]"""
import argparse
import collections
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=4)
    a = ap.parse_args()
    from stylegan2_step_time import Synthetic
    from diagan.models.stylegan2 import StyleGANDiscriminator, StyleGANGenerator
    from diagan.ops import conv as K
    from diagan.trainer import stylegan2 as TR
    dev = torch.device("cuda", 0)
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
    loader = torch.utils.data.DataLoader(ds, batch_size=a.batch, shuffle=True, drop_last=True)
    tr = TR.StyleGAN2Trainer(args, loader, G, D, g_optim, d_optim, g_ema, dev, "/tmp/sg2_time")
    zero = torch.tensor(0.0, device=dev)
    tr.r1_loss, tr.path_loss, tr.path_lengths = zero, zero, zero
    for i in range(1, 3):
        tr.train_step(i)
    torch.cuda.synchronize()
    table = collections.defaultdict(lambda: [0, 0.0, 0.0])

    def timed(name, fn, flop):
        def wrapper(geom, t, *args, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(geom, t, *args, **kw)
            e1.record()
            e1.synchronize()
            key = (name, geom.kind, f"{geom.R}x{geom.S}", geom.stride, geom.pad, geom.Ci, geom.Co, tuple(t.shape[:3]))
            row = table[key]
            row[0] += 1
            row[1] += e0.elapsed_time(e1)
            row[2] += flop(geom, t)
            return out
        return wrapper

    mm = lambda g, t: 2.0 * t.shape[0] * t.shape[1] * t.shape[2] * g.R * g.S * g.Ci * g.Co
    K.conv_fwd = timed("fwd", K.conv_fwd, lambda g, x: mm(g, x) * ((g.stride ** 2) if g.kind == "convT" else 1.0 / g.stride ** 2))
    K.conv_dgrad = timed("dgrad", K.conv_dgrad, mm)
    K.conv_wgrad = timed("wgrad", K.conv_wgrad, mm)
    for i in range(3, 3 + a.iters):
        tr.train_step(i)
    torch.cuda.synchronize()
    tot = sum(r[1] for r in table.values())
    print(f"{a.iters} iterations (numbers {3}..{2 + a.iters}): {tot / a.iters:.1f} ms of timed convolution launches per iteration")
    print(f"{'op':6s} {'kind':5s} {'RxS':4s} {'s':>1s} {'p':>1s} {'Ci':>4s} {'Co':>4s} {'B,H,W of the operand':22s} {'calls/it':>8s} {'ms/it':>8s} {'us/call':>8s} {'TFLOP/s':>8s}")
    for key, (n, ms, fl) in sorted(table.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{key[0]:6s} {key[1]:5s} {key[2]:4s} {key[3]:1d} {key[4]:1d} {key[5]:4d} {key[6]:4d} {str(key[7]):22s} {n / a.iters:8.1f} {ms / a.iters:8.2f} "
              f"{ms / n * 1e3:8.1f} {fl / ms / 1e9:8.1f}")


if __name__ == "__main__":
    main()
