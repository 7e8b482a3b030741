#!/usr/bin/env python
"""Micro-benchmark of the conv GEMM kernels on the SNGAN layer shapes (GPU box).
Prints TFLOP/s (algorithmic 2*M*N*K) and the fraction of the 157.3 TFLOP/s fp32 MFMA peak."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C

PEAK = 157.3e12
SHAPES = [
    # name, B, H, W, Ci, Co, R
    ("G32.b4.c1", 64, 32, 32, 256, 256, 3),
    ("G32.b3.c1", 64, 16, 16, 256, 256, 3),
    ("D32.b1.c2", 64, 32, 32, 128, 128, 3),
    ("D32.b2.c1", 64, 16, 16, 128, 128, 3),
    ("G64.b2.c1", 64, 8, 8, 1024, 512, 3),
    ("G64.b3.c1", 64, 16, 16, 512, 256, 3),
    ("G64.b4.c1", 64, 32, 32, 256, 128, 3),
    ("G64.b5.c1", 64, 64, 64, 128, 64, 3),
    ("G64.b5.c2", 64, 64, 64, 64, 64, 3),
    ("D64.b1.c2", 64, 64, 64, 64, 64, 3),
    ("D64.b2.c2", 64, 32, 32, 64, 128, 3),
    ("D64.b5.c2", 64, 4, 4, 512, 1024, 3),
    ("D64.b3.sc", 64, 16, 16, 128, 256, 1),
    ("D32.b3.c1", 64, 8, 8, 128, 128, 3),
    ("G32.b2.c1", 64, 8, 8, 256, 256, 3),
    ("G32.b2.sc", 64, 8, 8, 256, 256, 1),
    ("G32.b4.sc", 64, 32, 32, 256, 256, 1),
    ("D32.b2.sc", 64, 16, 16, 128, 128, 1),
    ("D32.b3.pair", 128, 8, 8, 128, 128, 3),
    ("D64.b5.pair", 128, 4, 4, 512, 1024, 3),
    ("D64.b4.pair", 128, 8, 8, 256, 512, 3),
    ("D32.b1.c1.pair", 128, 32, 32, 4, 128, 3),
    ("D64.b1.c1.pair", 128, 64, 64, 4, 64, 3),
    ("D32.b2.sc.pair", 128, 16, 16, 128, 128, 1),
    ("G32.b4.sc.lo", 64, 16, 16, 256, 256, 1),
]


def timeit(f, iters=20, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    only = sys.argv[1:] or None
    cfgs = [0]
    if os.environ.get("SWEEP"):
        cfgs = [1, 3]
    for name, B, H, W, Ci, Co, R in SHAPES:
        if only and name not in only:
            continue
        geom = C.Geom("conv", Ci, Co, R, R, 1, R // 2)
        x = torch.randn(B, H, W, Ci, device="cuda")
        dy = torch.randn(B, H, W, Co, device="cuda")
        wp = torch.randn(Co, geom.Kp, device="cuda") * 0.05
        wd = torch.zeros(Ci, geom.Kd, device="cuda")
        C.pack_weights(wp, Co, Ci, R * R, geom.Kp, geom.Kd, Wd=wd)
        grad = torch.zeros_like(wp)
        y = torch.empty(B, H, W, Co, device="cuda")
        dx = torch.empty(B, H, W, Ci, device="cuda")
        flop = 2.0 * B * H * W * Co * R * R * Ci
        for cfg in cfgs:
            tf = timeit(lambda: C.conv_fwd(geom, x, wp, out=y, tile_cfg=cfg))
            td = timeit(lambda: C.conv_dgrad(geom, dy, wd, (H, W), out=dx, tile_cfg=cfg))
            tw = timeit(lambda: C.conv_wgrad(geom, dy, x, grad, accumulate=False))
            print(f"{name:10s} cfg{cfg} M={B*H*W:6d} N={Co:4d} K={R*R*Ci:5d} {flop/1e9:7.1f} GF | "
                  f"fwd {tf*1e6:8.1f}us {flop/tf/1e12:6.1f}TF {flop/tf/PEAK:5.1%} | "
                  f"dgrad {td*1e6:8.1f}us {flop/td/1e12:6.1f}TF {flop/td/PEAK:5.1%} | "
                  f"wgrad {tw*1e6:8.1f}us {flop/tw/1e12:6.1f}TF {flop/tw/PEAK:5.1%}", flush=True)


if __name__ == "__main__":
    main()
