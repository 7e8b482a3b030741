#!/usr/bin/env python
"""Round 6: launch times of the one-launch modulation linear / demodulation (csrc/stylegan_dense.hip) against the general path.
    python tools/probe/dense_time.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch  # noqa: E402


def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    from diagan.models.op import fused_tail as FT
    from diagan.ops import diffconv as dc
    dev = torch.device("cuda", 0)
    for (B, K, C) in [(32, 512, 512), (32, 512, 128)]:
        x, W, b = torch.randn(B, K, device=dev, requires_grad=True), torch.randn(C, K, device=dev, requires_grad=True), torch.randn(C, device=dev, requires_grad=True)
        g = torch.randn(B, C, device=dev)

        def fused():
            y = FT.mod_linear(x, W, b, 0.04, 1.0)
            torch.autograd.grad(y, [x, W, b], g)

        def general():
            y = dc.linear(x, W, scale=0.04) + b
            torch.autograd.grad(y, [x, W, b], g)
        with torch.no_grad():
            f0, g0 = timed(lambda: FT.mod_linear(x, W, b, 0.04, 1.0)), timed(lambda: dc.linear(x, W, scale=0.04) + b)
        print(f"linear {B}x{K}->{C}: fwd one launch {f0:7.1f} us, general {g0:7.1f} us;  fwd+bwd {timed(fused):7.1f} / {timed(general):7.1f} us")
    for (B, Ci, Co) in [(32, 512, 512), (32, 256, 128)]:
        s, w = torch.randn(B, Ci, device=dev, requires_grad=True), torch.randn(Co, Ci, 3, 3, device=dev, requires_grad=True)
        g = torch.randn(B, Co, device=dev)
        sc2 = 1.0 / (Ci * 9)

        def fused():
            torch.autograd.grad(FT.demod(s, w, sc2, 1e-8), [s, w], g)

        def general():
            torch.autograd.grad(torch.rsqrt(dc.linear(s.square(), w.square().sum((2, 3)), scale=sc2) + 1e-8), [s, w], g)
        with torch.no_grad():
            f0 = timed(lambda: FT.demod(s, w, sc2, 1e-8))
            g0 = timed(lambda: torch.rsqrt(dc.linear(s.square(), w.square().sum((2, 3)), scale=sc2) + 1e-8))
        print(f"demod {B}x{Ci}->{Co}: fwd one launch {f0:7.1f} us, general {g0:7.1f} us;  fwd+bwd {timed(fused):7.1f} / {timed(general):7.1f} us")


if __name__ == "__main__":
    main()
