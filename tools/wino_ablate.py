#!/usr/bin/env python
"""Diagnostic (GPU box, library built with -DDIAGAN_WINO_ABLATE): launch time of the Winograd forward kernel with parts
of its K loop switched off through the tune bits (results are garbage then) -- what each part costs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan import _native as nat
from diagan.ops import conv as C

BITS = {16: "no transform", 32: "no input loads", 64: "no weight DMA", 128: "no barrier", 256: "no MFMA", 512: "no epilogue"}


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    dev = "cuda"
    for B, H, W, Ci, Co in ((64, 32, 32, 256, 256), (128, 64, 64, 64, 64)):
        geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
        x = torch.randn(B, H, W, Ci, device=dev)
        wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
        f = lambda: C.conv_fwd(geom, x, wp, tile_cfg=9)
        combos = [0, 16, 32, 64, 128, 256, 512, 16 | 32, 16 | 32 | 64, 16 | 32 | 64 | 128, 256 | 16 | 32, 256 | 16 | 32 | 64,
                  16 | 32 | 64 | 512, 16 | 32 | 64 | 128 | 512, 16 | 32 | 64 | 128 | 256 | 512]
        mf = 2.0 * B * H * W * Co * 4 * Ci / 157.3e12 * 1e6
        print(f"B={B} {H}x{W} Ci={Ci} Co={Co}: MFMA-only time at peak {mf:.1f} us")
        for c in combos:
            nat.call("diagan_conv_gemm_tune", 0, c, 0)
            t = timeit(f)
            names = ", ".join(n for b, n in BITS.items() if c & b) or "full kernel"
            print(f"  tune {c:4d}  {t:8.1f} us   {names}", flush=True)
        nat.call("diagan_conv_gemm_tune", 0, -1, 0)


if __name__ == "__main__":
    main()
