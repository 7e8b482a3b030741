#!/usr/bin/env python
"""Diagnostic (GPU box): the Winograd forward kernel's instruction-schedule variants (tune bits 12-13) side by side."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan import _native as nat
from diagan.ops import conv as C
from wino_ablate import timeit


def main():
    dev = "cuda"
    for B, H, W, Ci, Co in ((64, 32, 32, 256, 256), (128, 64, 64, 64, 64), (128, 16, 16, 128, 256), (320, 32, 32, 256, 128)):
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, W, Ci, device=dev)
        wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
        sc, sh = torch.rand(Ci, device=dev) + 0.5, torch.randn(Ci, device=dev) * 0.3
        for pname, pro in (("plain", None), ("bn+relu", (C.PRO_AFFINE_RELU, sc, sh))):
            f = lambda cfg=9: C.conv_fwd(geom, x, wp, pro=pro, tile_cfg=cfg)
            ref = f(7)
            line = f"B={B} {H}x{W} Ci={Ci} Co={Co} {pname:8s}"
            for sched in (0, 1, 2, 0, 1, 2):
                nat.call("diagan_conv_gemm_tune", 0, sched << 12, 0)
                err = ((f() - ref).abs().max() / ref.abs().max()).item()
                t = timeit(f)
                line += f" | s{sched} {t:7.1f} us ({err:.0e})"
            nat.call("diagan_conv_gemm_tune", 0, -1, 0)
            t = timeit(lambda: f(10))
            line += f" | staged {t:7.1f}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
