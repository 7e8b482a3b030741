#!/usr/bin/env python
"""The discriminators' first convolution (RGB padded to 4 channels -> 128 / 64, 3x3, K = 36): time of every implicit-GEMM tile
against the output stream's HBM time (GPU box; timing only)."""
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

for B, H, Co in ((128, 32, 128), (64, 32, 128), (128, 64, 64), (64, 64, 64)):
    geom = C.Geom("conv", 4, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, H, 4, device="cuda")
    wp = torch.randn(Co, geom.Kp, device="cuda")
    bias = torch.randn(Co, device="cuda")
    out_mb = B * H * H * Co * 4 / 1e6
    row = [f"B={B} {H}x{H} Co={Co} Kp={geom.Kp}: output {out_mb:.0f} MB = {out_mb / 8e6 * 1e6:.1f} us at 8 TB/s |"]
    for cfg in (0, 1, 3, 5, 7, 8):
        try:
            row.append(f"cfg{cfg} {t_us(lambda: C.conv_fwd(geom, x, wp, bias=bias, tile_cfg=cfg)):.1f}")
        except Exception as e:
            row.append(f"cfg{cfg} n/a")
    print(" ".join(row))
    os.environ  # (cfg0 above is the dedicated kernel when DIAGAN_CONV_CI4 != 0)
