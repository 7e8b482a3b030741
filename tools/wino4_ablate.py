#!/usr/bin/env python
"""Ablation of the F(4x4,3x3) kernel's K loop (diagnostic build: make EXTRA=-DDIAGAN_WINO_ABLATE): launch time with parts of
the loop switched off through ConvGemmArgs::tune (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "self-diagnosing-gan_amd"))
import torch
from diagan.ops import conv as C
from diagan import _native as nat

def timeit(f, iters=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

dev = "cuda"
cases = {"full": 0, "no row+col pass": 16, "no col pass": 4096, "no input loads": 32, "no loads, no transform": 48, "no DMA": 64,
         "no barrier": 128, "no MFMA": 256, "no fragment reads": 2048, "MFMA only": 16 | 32 | 64 | 128 | 2048,
         "MFMA + fragments": 16 | 32 | 64 | 128, "MFMA + fragments + barrier": 16 | 32 | 64,
         "MFMA+frag+DMA+barrier": 16 | 32, "all but MFMA": 256, "transform only": 256 | 2048 | 64,
         "no DMA no frag": 64 | 2048}
for B, H, W, Ci, Co in ((64, 32, 32, 256, 256), (384, 32, 32, 256, 256)):
    geom = C.Geom("conv", Ci, Co, 3, 3, 1, 1)
    x = torch.randn(B, H, W, Ci, device=dev)
    wp = torch.randn(Co, geom.Kp, device=dev) * (9 * Ci) ** -0.5
    nk = Ci // 8
    wgs_per_cu = (B * H * W // 512) * (Co // 64) / 256
    for name, bits in cases.items():
        nat.call("diagan_conv_gemm_tune", 0, bits, 0)
        t = timeit(lambda: C.conv_fwd(geom, x, wp, tile_cfg=13))
        cyc = t / (wgs_per_cu * nk) * 2.2e9
        print(f"B={B} {H}x{W} {Ci}->{Co} {name:28s} {t*1e6:8.1f} us   ~{cyc:7.0f} cycles per K-step (2.2 GHz, {wgs_per_cu:.0f} rounds)", flush=True)
    nat.call("diagan_conv_gemm_tune", 0, -1, 0)
