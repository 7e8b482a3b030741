#!/usr/bin/env python
"""Launch-size policy check (GPU box): for the few-tile 3x3 layers of the SNGAN networks, the Winograd kernel at split-K
1..4 (forced), the implicit GEMM, and what tile_cfg 0 picks -- forward and data-gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan import _native as nat
from diagan.ops import conv as C
from wino_check import timeit

SHAPES = [(64, 4, 4, 1024, 1024), (128, 4, 4, 1024, 1024), (128, 4, 4, 512, 1024), (64, 8, 8, 512, 512), (128, 8, 8, 512, 512),
          (128, 8, 8, 256, 512), (64, 8, 8, 1024, 512), (64, 16, 16, 512, 256), (64, 16, 16, 256, 256), (128, 16, 16, 256, 256),
          (128, 16, 16, 128, 256), (64, 8, 8, 256, 256), (128, 8, 8, 128, 128), (64, 32, 32, 128, 128), (64, 16, 16, 128, 128),
          (64, 32, 32, 256, 128), (64, 64, 64, 128, 64), (64, 64, 64, 64, 64)]


def main():
    dev = "cuda"
    for B, H, W, Ci, Co in SHAPES:
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, W, Ci, device=dev)
        wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
        f = lambda cfg: C.conv_fwd(geom, x, wp, pro=(C.PRO_RELU, None, None), tile_cfg=cfg)
        wgs = -(-B * H * W // 4 // 64) * -(-Co // 64)
        line = f"B={B:3d} {H:2d}x{W:2d} Ci={Ci:4d} Co={Co:4d} wgs {wgs:4d}:"
        best = None
        for ks in (1, 2, 3, 4):
            nat.call("diagan_conv_gemm_tune", ks, -1, 0)
            t = timeit(lambda: f(9)) * 1e6
            line += f" ks{ks} {t:6.1f}"
            best = min(best or t, t)
        nat.call("diagan_conv_gemm_tune", 0, -1, 0)
        C.set_winograd(False)
        ti = timeit(lambda: f(0)) * 1e6
        C.set_winograd(None)
        ta = timeit(lambda: f(0)) * 1e6
        best = min(best, ti)
        print(line + f" | implicit {ti:6.1f} | auto {ta:6.1f}{'   <-- ' + format(ta / best, '.2f') + 'x of best' if ta > 1.05 * best else ''}", flush=True)


if __name__ == "__main__":
    main()
