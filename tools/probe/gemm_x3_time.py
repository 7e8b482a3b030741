#!/usr/bin/env python
"""The split-operand implicit GEMM (tile_cfg 16) against the fp32 lone-tile kernel (14) and the plain 64x64 tile (7) on the
discriminator's 8x8 layers (GPU box; weights pre-split once, as in a training step: the hint mechanism)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C

def t_us(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B, H, Ci, Co in ((128, 8, 128, 128), (64, 8, 128, 128), (64, 4, 256, 256), (256, 8, 128, 128)):
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, H, Ci, device="cuda")
    wp = torch.randn(Co, geom.Kp, device="cuda") * (9 * Ci) ** -0.5
    bias = torch.randn(Co, device="cuda")
    res = torch.randn(B, H, H, Co, device="cuda")
    out = torch.empty(B, H, H, Co, device="cuda")
    relu = (C.PRO_RELU, None, None)
    batch = C.WinoWeightBatch()
    site = batch.site(lambda: wp, Co, Ci, geom.Kp)
    row = [f"B={B} {H}x{H} {Ci}->{Co} (M={B*H*H}):"]
    for cfg in (7, 14, 16):
        def f():
            if cfg == 16:
                batch.prepare(1)
            return C.conv_fwd(geom, x, wp, bias=bias, residual=res, res_relu=True, pro=relu, out=out, tile_cfg=cfg,
                              wsite=site if cfg == 16 else None, wversion=1)
        row.append(f"cfg{cfg} {t_us(f):.1f} us")
    print(" ".join(row))
