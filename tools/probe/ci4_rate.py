#!/usr/bin/env python
"""conv3x3_ci4 against plain fills of the same output size (what does a pure write stream reach?), and its time against the batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C

def t_us(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B, H, Co in ((16, 32, 128), (32, 32, 128), (64, 32, 128), (128, 32, 128), (256, 32, 128), (512, 32, 128), (128, 64, 64), (512, 64, 64)):
    geom = C.Geom("conv", 4, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, H, 4, device="cuda")
    wp = torch.randn(Co, geom.Kp, device="cuda")
    bias = torch.randn(Co, device="cuda")
    out = torch.empty(B, H, H, Co, device="cuda")
    mb = out.numel() * 4 / 1e6
    a = t_us(lambda: C.conv_fwd(geom, x, wp, bias=bias, out=out))
    b = t_us(lambda: C.conv_fwd(geom, x, wp, bias=bias, out=out, tile_cfg=7))
    f = t_us(lambda: out.fill_(1.0))
    c = t_us(lambda: out.copy_(out))  if False else 0
    print(f"B={B} {H}x{H} Co={Co}: {mb:.0f} MB | ci4 {a:.1f} us ({mb / a:.2f} TB/s... {mb / a / 1e0 * 1e-0:.2f}) | gemm {b:.1f} | fill_ {f:.1f} us ({mb / f * 1e-0:.2f} MB/us)")
