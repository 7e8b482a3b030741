#!/usr/bin/env python
"""Diagnostic (GPU box, library built with -DDIAGAN_WINO_ABLATE): where a K-step of the Winograd forward kernel goes --
per-wave s_memtime stamps summed over the K loop (tune bit 10), median over all waves of all workgroups."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import numpy as np
import torch
from diagan import _native as nat
from diagan.ops import conv as C

PHASES = ["DMA / load issue + fragment reads", "first 16 MFMAs (issue)", "wait: input loads", "second 16 MFMAs + transform",
          "wait: weight DMA", "barrier", "set-up + first stage", "total to the epilogue"]


def main():
    dev = "cuda"
    slots = 1 << 17
    buf = torch.zeros(slots * 8, dtype=torch.int64, device=dev)
    for B, H, W, Ci, Co in ((64, 32, 32, 256, 256), (128, 64, 64, 64, 64)):
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, W, Ci, device=dev)
        wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
        f = lambda: C.conv_fwd(geom, x, wp, tile_cfg=9)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        nwg = (B * H * W // 256) * (Co // 64)
        nk = Ci // 8
        for extra in (0, 16, 32, 16 | 32):
            buf.zero_()
            nat.call("diagan_conv_gemm_set_stamp_buffer", buf.data_ptr(), slots)
            nat.call("diagan_conv_gemm_tune", 0, 1024 | extra, 0)
            f()
            torch.cuda.synchronize()
            nat.call("diagan_conv_gemm_set_stamp_buffer", None, 0)
            nat.call("diagan_conv_gemm_tune", 0, -1, 0)
            t = buf[: nwg * 64].cpu().numpy().reshape(nwg, 8, 8).astype(np.float64)
            print(f"B={B} {H}x{W} Ci={Ci} Co={Co}: {nwg} workgroups, {nk} K-steps, ablation bits {extra} "
                  f"(16 = no transform, 32 = no input loads); cycles per K-step (MFMA-bound: 4096 per SIMD = 2 waves x 32 x 64)")
            for i, name in enumerate(PHASES):
                v = t[:, :, i].reshape(-1)
                div = nk if i < 6 else 1
                print(f"   {name:36s} median {np.median(v) / div:9.0f}   p10 {np.percentile(v, 10) / div:9.0f}   "
                      f"p90 {np.percentile(v, 90) / div:9.0f}")
            loop = t[:, :, :6].sum(2).reshape(-1)
            print(f"   {'K loop, per step':36s} median {np.median(loop) / nk:9.0f}")


if __name__ == "__main__":
    main()
